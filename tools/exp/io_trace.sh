#!/bin/bash
# GPU box: kernel trace of the files-to-files pipeline: busy fraction and per-kernel time per frame in the steady window
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/io_trace; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o r -- python3 bench.py --io-only --io-simple --steps 40 --writers 12 $BENCH_EXTRA > $OUT/line.json 2> $OUT/err.log || exit 1
python3 tools/trace_busy.py $OUT/kt z3:: k_fp_tile k_fp_ k_funpack > $OUT/busy.txt
python3 - <<PY >> $OUT/busy.txt
import csv, glob
f = (glob.glob("$OUT/kt/*/*kernel_trace.csv") + glob.glob("$OUT/kt/*kernel_trace.csv"))[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
marks = sorted(e for s_, e, n in rows if 'k_final_rows' in n)
a, b = marks[-40], marks[-10]
tot = {}
for s, e, n in rows:
    if e <= a or s >= b: continue
    k = n.split('(')[0].split('<')[0][-40:]
    t = tot.setdefault(k, [0, 0]); t[0] += min(e, b) - max(s, a); t[1] += 1
print('per frame (30 frames), top kernels by time:')
for k, (t, c) in sorted(tot.items(), key=lambda x: -x[1][0])[:28]:
    print('%-42s %7.3f ms  %6.1f launches' % (k, t / 30e6, c / 30))
print('sum of all kernel durations per frame: %.2f ms' % (sum(t for t, c in tot.values()) / 30e6))
PY
rm -rf $OUT/kt
cat $OUT/busy.txt; cat $OUT/line.json | head -c 600
