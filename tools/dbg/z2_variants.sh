#!/bin/bash
# GPU box: time bbx_zogy_frame for builds of $SRC.hip (bbx_zogy2 or bbx_zogy3) with different -D flags (scratch copies of
# (needs the object files of the other sources on the box: take the `*.o` line out of .gpurunignore for such a run)
# the library under /tmp; the product .so is not touched).  PROF=1: per-kernel rocprofv3 stats instead.
SRC=${SRC:-bbx_zogy3}
FL="--offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-function"
mkdir -p /tmp/z2v
export TMPDIR=/tmp
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  rm -rf /tmp/z2v/$name; mkdir -p /tmp/z2v/$name/blackbox_amd /tmp/z2v/$name/tools/dbg
  cp -r blackbox_amd/*.py blackbox_amd/libbbx_host.so /tmp/z2v/$name/blackbox_amd/
  /opt/rocm/bin/hipcc $FL $defs -c blackbox_amd/csrc/$SRC.hip -o /tmp/z2v/$name/z2.o || exit 1
  objs=$(ls blackbox_amd/csrc/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/z2v/$name/blackbox_amd/libbbx_hip.so $objs /tmp/z2v/$name/z2.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib || exit 1
  cp bench.py /tmp/z2v/$name/; cp tools/dbg/z2_time.py /tmp/z2v/$name/tools/dbg/; cp tools/prof_summary.py /tmp/z2v/$name/tools/
  echo "== $name ($defs)"
  if [ -n "$PROF" ]; then
    (cd /tmp/z2v/$name && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/z2v/$name/prof -o r -- python3 tools/dbg/z2_time.py > /tmp/z2v/$name/prof.log 2>&1; python3 tools/prof_summary.py /tmp/z2v/$name/prof 7 50 | grep "z[23]::\|total" | cut -c1-40,70-130)
  else
    (cd /tmp/z2v/$name && timeout -k 10 120 python3 tools/dbg/z2_time.py 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -2)
  fi
done
