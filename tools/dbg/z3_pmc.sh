#!/bin/bash
# GPU box: SQ counters of the bbx_zogy_frame kernels (LDS bank conflicts, VALU / LDS busy) -> gpurun_out/z3_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/z3_pmc; rm -rf $OUT; mkdir -p $OUT
SETS=("SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES")
[ -n "$FULL" ] && SETS+=("SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY")
for set in "${SETS[@]}"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  N=2 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o r -- python3 tools/dbg/z3_time.py > $OUT/$tag.log 2>&1 || { tail -5 $OUT/$tag.log; }
done
python3 - <<'PY' > gpurun_out/z3_pmc.txt
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('gpurun_out/z3_pmc/*/*counter_collection.csv') + glob.glob('gpurun_out/z3_pmc/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').strip()
        if 'z3::' not in name: continue
        a = acc[name][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in sorted(acc.items()):
    print(k)
    for c, (n, v) in sorted(d.items()):
        print('   %-24s %14.0f  (n=%d)' % (c, v / n, n))
PY
cat gpurun_out/z3_pmc.txt
