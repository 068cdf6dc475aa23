#!/bin/bash
# GPU box: durations of the kernels matching PATTERN in one serial frame, product library and scratch libraries
# usage: tools/exp/seq_vars.sh PATTERN name [name ...]
cd "$GRAFT_REPO_ROOT" || exit 1
pat=$1; shift
for v in product "$@"; do
  if [ "$v" = product ]; then unset BBX_LIB_PATH; else export BBX_LIB_PATH=tools/exp/_var/$v/libbbx_hip.so; fi
  bash tools/exp/frame_seq.sh > /dev/null || exit 1
  echo "== $v"; grep "$pat" gpurun_out/frame_seq.txt
done
