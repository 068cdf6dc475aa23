#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/bkp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bkp -o r -- python3 tools/exp/bkg_prof.py > gpurun_out/bkp.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
f = (glob.glob('gpurun_out/bkp/*/*_kernel_stats.csv') + glob.glob('gpurun_out/bkp/*_kernel_stats.csv'))[0]
rows = [r for r in csv.DictReader(open(f)) if int(r['Calls']) in (4, 8, 12, 16)]
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:12]:
    print(r['Name'][:70], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1))
PY
rm -rf gpurun_out/bkp
