#!/bin/bash
# GPU box: tools/dbg/box_time.py with the product library and with scratch builds (tools/exp/libvar.sh NAME flags bbx_bkg.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== product"; python3 tools/dbg/box_time.py 2>&1 | grep "full_sort\|medians"
for v in "$@"; do echo "== $v"; BBX_LIB_PATH=tools/exp/_var/$v/libbbx_hip.so python3 tools/dbg/box_time.py 2>&1 | grep "full_sort=0\|medians"; done
