#!/bin/bash
# scratch build of the library with compile-time switches in some of its sources, under tools/exp/_var/<name>/ (git-ignored,
# travels with gpurun; BBX_LIB_PATH selects it); the product library is never replaced.
# usage: tools/exp/libvar.sh NAME "-DX -DY" file.hip [file.hip ...]
set -e
cd "$(dirname "$0")/../.."
name=$1; flags=$2; shift 2
d=tools/exp/_var/$name; mkdir -p $d; rm -f $d/*.o
skip=""
for f in "$@"; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-function $flags -c blackbox_amd/csrc/$b.hip -o $d/$b.o
  skip="$skip -e /$b.o"
done
objs=$(ls blackbox_amd/csrc/*.o | grep -v $skip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libbbx_hip.so $d/*.o $objs -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
echo built $d
