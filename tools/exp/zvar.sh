#!/bin/bash
# Build container: scratch builds of the library with -D variants of one HIP source (default bbx_zogy3), for timing on
# the GPU box with BBX_LIB_PATH (the product .so is not touched).
#   tools/exp/zvar.sh name1:"-DFOO -DBAR=2" name2:"" ...   ->  tools/exp/_var/<name>/libbbx_hip.so
SRC=${SRC:-bbx_zogy3}
FL="--offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-function"
cd "$(dirname "$0")/../.." || exit 1
make -j8 all > /dev/null || exit 1
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  mkdir -p tools/exp/_var/$name
  /opt/rocm/bin/hipcc $FL $defs -c blackbox_amd/csrc/$SRC.hip -o tools/exp/_var/$name/v.o || exit 1
  objs=$(ls blackbox_amd/csrc/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_var/$name/libbbx_hip.so $objs tools/exp/_var/$name/v.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib || exit 1
  rm -f tools/exp/_var/$name/v.o
  echo "built $name ($defs)"
done
