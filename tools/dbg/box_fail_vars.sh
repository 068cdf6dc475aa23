#!/bin/bash
# GPU box: boxes left to the full sort for scratch builds with -DBOXV (tools/dbg/box_fail.py)
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do echo "== $v"; BBX_LIB_PATH=tools/exp/_var/$v/libbbx_hip.so BBX_DBG_BOX_NOFALLBACK=1 python3 tools/dbg/box_fail.py 2>&1 | grep "boxes\|reason"; done
