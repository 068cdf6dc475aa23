#!/bin/bash
# GPU box: time bbx_zogy_frame's kernels for the product library and the scratch builds named on the command line
cd "$GRAFT_REPO_ROOT" || exit 1
N=${N:-8} timeout -k 10 200 python3 tools/dbg/z3_time.py || exit 1
for v in "$@"; do
  BBX_LIB_PATH=tools/exp/_var/$v/libbbx_hip.so N=${N:-8} timeout -k 10 200 python3 tools/dbg/z3_time.py || exit 1
done
