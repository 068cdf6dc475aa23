#!/bin/bash
# GPU box: kernel trace of the child `python blackbox.py --image_list` of bench.py's cli_image_list: how busy the GPU is in
# the steady part of the list and which kernels fill a frame's time -> gpurun_out/cli_trace/busy.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/cli_trace; rm -rf $OUT; mkdir -p $OUT
BBX_CLI_TRACE=$OUT/kt BBX_CLI_NO_POOL=1 timeout -k 10 700 python3 bench.py --proc-only --steps ${NFILES:-96} > $OUT/line.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 tools/trace_busy.py $OUT/kt z3:: k_fp_tile k_fp_ k_funpack > $OUT/busy.txt
python3 - <<PY >> $OUT/busy.txt
import csv, glob
f = (glob.glob("$OUT/kt/*/*kernel_trace.csv") + glob.glob("$OUT/kt/*kernel_trace.csv"))[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
marks = sorted(e for s_, e, n in rows if 'k_final_rows' in n)
import os
W0, W1 = int(os.environ.get('W0', 40)), int(os.environ.get('W1', 10))
NW = W0 - W1
a, b = marks[-W0], marks[-W1]
tot = {}
for s, e, n in rows:
    if e <= a or s >= b: continue
    k = n.split('(')[0].split('<')[0][-40:]
    t = tot.setdefault(k, [0, 0]); t[0] += min(e, b) - max(s, a); t[1] += 1
print('per frame (%d frames, %.2f ms per frame in the window), top kernels by time:' % (NW, (b - a) / 1e6 / NW))
for k, (t, c) in sorted(tot.items(), key=lambda x: -x[1][0])[:int(os.environ.get('NTOP', 32))]:
    print('%-42s %7.3f ms  %6.1f launches' % (k, t / (NW * 1e6), c / NW))
fp = [(s_, e, n) for s_, e, n in rows if 'k_fp_tile' in n and a <= s_ < b]
fp.sort()
print('k_fp_tile launches of two frames in the window, in order (template arguments = bytes per pixel, float input, mode; us):')
print('   ' + '  '.join('%s %.0f' % (n.split('k_fp_tile')[1].split('(')[0], (e - s_) / 1e3) for s_, e, n in fp[:24]))
print('sum of all kernel durations per frame: %.2f ms' % (sum(t for t, c in tot.values()) / (NW * 1e6)))
PY
rm -rf $OUT/kt
cat $OUT/busy.txt; python3 tools/dbg/bench_sum.py $OUT/line.json 2>/dev/null | tail -5; grep -o '"image_list": {[^}]*}' $OUT/line.json | head -c 700
