#!/bin/bash
# GPU box: tools/dbg/fp_time.py with the product library and with scratch builds (tools/exp/fpvar.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== product"; python3 tools/dbg/fp_time.py 2>&1 | grep "hist_only"
for v in "$@"; do echo "== $v"; BBX_LIB_PATH=tools/exp/_var/$v/libbbx_hip.so python3 tools/dbg/fp_time.py 2>&1 | grep "tile_only=0"; done
