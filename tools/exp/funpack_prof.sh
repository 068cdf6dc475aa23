#!/bin/bash
# GPU box: kernel time of k_funpack on a full raw frame (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/fup
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fup -o r -- python3 tools/exp/funpack_time.py > gpurun_out/fup.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
f = (glob.glob('gpurun_out/fup/*/*_kernel_stats.csv') + glob.glob('gpurun_out/fup/*_kernel_stats.csv'))[0]
for r in csv.DictReader(open(f)):
    if 'funpack' in r['Name'] or 'k_fp_' in r['Name']:
        print(r['Name'][:60], 'calls', r['Calls'], 'avg_us', float(r['AverageNs']) / 1e3)
PY
rm -rf gpurun_out/fup
