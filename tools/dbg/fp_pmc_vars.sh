#!/bin/bash
# GPU box: vector / scalar / LDS instruction counts of k_fp_tile<4, true, 1> for scratch builds with parts compiled out (tools/exp/fpvar.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/fp_pmc_vars.txt
for v in product "$@"; do
  if [ "$v" = product ]; then unset BBX_LIB_PATH; else export BBX_LIB_PATH=$GRAFT_REPO_ROOT/tools/exp/_var/$v/libbbx_hip.so; fi
  OUT=gpurun_out/fpv; rm -rf $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT -o r -- python3 tools/dbg/fp_time.py > gpurun_out/fpv.log 2>&1 || { tail -3 gpurun_out/fpv.log; }
  python3 - "$v" <<'PY' >> gpurun_out/fp_pmc_vars.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('gpurun_out/fpv/*counter_collection.csv') + glob.glob('gpurun_out/fpv/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_fp_tile<4, true, 1>' in r['Kernel_Name'] or 'k_fp_tileILi4ELb1ELi1' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
npx = 10560 * 10560
print('%-10s' % sys.argv[1], '  '.join('%s %.0f M = %.0f per pixel' % (c[9:], v / n / 1e6, v / n * 64 / npx) for c, (n, v) in sorted(acc.items())))
PY
  rm -rf $OUT
done
cat gpurun_out/fp_pmc_vars.txt
