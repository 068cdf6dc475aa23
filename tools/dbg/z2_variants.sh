#!/bin/bash
# GPU box: time bbx_zogy_frame for builds of $SRC.hip (bbx_zogy2 or bbx_zogy3) with different -D flags (scratch copies of
# (needs the object files of the other sources on the box: take the `*.o` line out of .gpurunignore for such a run)
# the library under /tmp; the product .so is not touched).  PROF=1: per-kernel rocprofv3 stats instead.
SRC=${SRC:-bbx_zogy3}
TS=${TS:-z2_time}            # timing script under tools/dbg
PAT=${PAT:-z[23]::}          # kernels to list
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
  cp bench.py /tmp/z2v/$name/; cp tools/dbg/$TS.py /tmp/z2v/$name/tools/dbg/; cp tools/prof_summary.py /tmp/z2v/$name/tools/
  echo "== $name ($defs)"
  if [ -n "$PMC" ]; then
    # effective clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration
    (cd /tmp/z2v/$name && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/z2v/$name/prof -o r -- python3 tools/dbg/$TS.py > /tmp/z2v/$name/prof.log 2>&1
     timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/z2v/$name/pmc -o r -- python3 tools/dbg/$TS.py > /tmp/z2v/$name/pmc.log 2>&1
     python3 - <<PY
import csv, glob, re
st = {re.sub(r'\(.*', '', r['Name']): float(r['AverageNs']) for r in csv.DictReader(open(glob.glob('/tmp/z2v/$name/prof/*kernel_stats.csv')[0]))}
acc = {}
for r in csv.DictReader(open(glob.glob('/tmp/z2v/$name/pmc/*counter_collection.csv')[0])):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    k = re.sub(r'\(.*', '', r['Kernel_Name']); a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in acc.items():
    if 'z3::' in k and k in st: print('%-40s avg %.0f us  clock %.2f GHz' % (k[:40], st[k] / 1e3, v / n / 8 / st[k]))
PY
    )
  elif [ -n "$PROF" ]; then
    (cd /tmp/z2v/$name && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/z2v/$name/prof -o r -- python3 tools/dbg/$TS.py > /tmp/z2v/$name/prof.log 2>&1; python3 tools/prof_summary.py /tmp/z2v/$name/prof 7 50 | grep "$PAT\|total" | cut -c1-40,70-130)
  else
    (cd /tmp/z2v/$name && timeout -k 10 120 python3 tools/dbg/$TS.py 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -2)
  fi
done
