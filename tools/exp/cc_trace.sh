#!/bin/bash
# GPU box: per-launch durations of selected kernels from a short traced bench run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/cctr
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cctr -o r -- python3 bench.py --steps 8 --warmup 2 --no-cpu --depth 1 > gpurun_out/cctr.log 2>&1 || exit 1
python3 - "$@" <<'PY'
import csv, sys, collections
pat = sys.argv[1:] or ['k_cc_union']
rows = list(csv.DictReader(open('gpurun_out/cctr/r_kernel_trace.csv')))
for p in pat:
    d = [ (int(r['Start_Timestamp']), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows if r['Kernel_Name'].startswith(p)]
    d.sort()
    print(p, ' '.join('%.1f' % x[1] for x in d[-24:]))
PY
rm -rf gpurun_out/cctr
