#!/bin/bash
# build variants of one source file with -D flags:  tools/exp/variants.sh <file.hip> name1:"-DX=1 -DY=2" name2:"..."
# -> tools/exp/variants/libbbx_<name>.so   (run on the GPU box with tools/exp/run_variants.sh)
src=$1; shift
mkdir -p tools/exp/variants
FL="--offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wall -Wno-unused-function"
base=$(basename $src .hip)
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  /opt/rocm/bin/hipcc $FL $defs -c $src -o tools/exp/variants/${base}_$name.o || exit 1
  objs=$(ls blackbox_amd/csrc/*.o | grep -v "/$base.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/variants/libbbx_$name.so $objs tools/exp/variants/${base}_$name.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib || exit 1
  rm tools/exp/variants/${base}_$name.o
  echo built $name
done
