#!/bin/bash
# GPU box: time of k_bkg_boxstats_fast with parts knocked out (scratch builds bk0.. of tools/exp/libvar.sh, no fallback launch)
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do echo "== $v"; BBX_DBG_BOX_NOFALLBACK=1 BBX_LIB_PATH=tools/exp/_var/$v/libbbx_hip.so python3 tools/dbg/box_time.py 2>&1 | grep "full_sort=0"; done
