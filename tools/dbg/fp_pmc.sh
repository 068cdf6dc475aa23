#!/bin/bash
# GPU box: SQ counters of k_fp_tile and k_bkg_boxstats_fast (instruction mix, busy / wait cycles) -> gpurun_out/fp_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/fp_pmc; rm -rf $OUT; mkdir -p $OUT
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES")
for set in "${SETS[@]}"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o r -- python3 tools/dbg/${PMC_PROG:-fp_time.py} > $OUT/$tag.log 2>&1 || { tail -5 $OUT/$tag.log; }
done
python3 - <<'PY' > gpurun_out/fp_pmc.txt
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('gpurun_out/fp_pmc/*/*counter_collection.csv') + glob.glob('gpurun_out/fp_pmc/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('void ', '').strip()[:60]
        if not ('k_fp_tile' in name or 'boxstats' in name): continue
        a = acc[name][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in sorted(acc.items()):
    print(k)
    for c, (n, v) in sorted(d.items()):
        print('   %-24s %16.0f  (n=%d)' % (c, v / n, n))
PY
rm -rf $OUT
cat gpurun_out/fp_pmc.txt
