#!/bin/bash
# GPU box: run "$@" once per variant library
cp blackbox_amd/libbbx_hip.so /tmp/libbbx_orig.so
for f in tools/exp/variants/libbbx_*.so; do
  name=$(basename $f .so); name=${name#libbbx_}
  cp $f blackbox_amd/libbbx_hip.so
  echo "== $name"; "$@" 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -3
done
cp /tmp/libbbx_orig.so blackbox_amd/libbbx_hip.so
