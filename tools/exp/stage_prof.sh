#!/bin/bash
# stage bench refresh + kernel breakdown of the same run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/stp
true
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stp -o r -- python3 tools/stage_bench.py --reps 3 > gpurun_out/stp/log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/stp/**/r_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:40]:
    print('%-90s calls=%6s tot_ms=%9.2f avg_us=%9.1f'%(r['Name'][:90],r['Calls'],float(r['TotalDurationNs'])/1e6,float(r['AverageNs'])/1e3))
PY
