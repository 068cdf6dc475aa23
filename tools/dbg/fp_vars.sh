#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python3 tools/dbg/fp_time.py 2>&1 | tail -1
for v in "$@"; do echo "$v: $(BBX_LIB_PATH=tools/exp/_var/$v/libbbx_hip.so python3 tools/dbg/fp_time.py 2>&1 | tail -1)"; done
