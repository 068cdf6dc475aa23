#!/bin/bash
# scratch builds of the library with knock-out switches of bbx_fpack.hip (FPV_*), under tools/exp/_var/<name>/ (git-ignored,
# travels with gpurun); the product library is never replaced.  usage: tools/exp/fpvar.sh NAME "-DFPV_X -DFPV_Y" ...
set -e
cd "$(dirname "$0")/../.."
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  d=tools/exp/_var/$name; mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-function $flags -c blackbox_amd/csrc/bbx_fpack.hip -o $d/bbx_fpack.o
  objs=$(ls blackbox_amd/csrc/*.o | grep -v bbx_fpack.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libbbx_hip.so $d/bbx_fpack.o $objs -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
  echo built $d
done
