#!/bin/bash
# GPU box: instruction-cache / fetch counters of the bbx_zogy_frame kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/z3_ic; rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*" $OUT/counters.txt | sort -u > $OUT/names.txt
cat $OUT/names.txt | tr '\n' ' '
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  N=2 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o r -- python3 tools/dbg/z3_time.py > $OUT/$tag.log 2>&1 || { tail -5 $OUT/$tag.log; }
done
python3 - <<'PY'
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('gpurun_out/z3_ic/*/*counter_collection.csv') + glob.glob('gpurun_out/z3_ic/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').strip()
        if 'z3::' not in name: continue
        a = acc[name][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in sorted(acc.items()):
    print(k)
    for c, (n, v) in sorted(d.items()):
        print('   %-24s %14.0f  (n=%d)' % (c, v / n, n))
PY
