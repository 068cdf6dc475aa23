#!/bin/bash
# GPU box: memory-copy trace + kernel trace of the files-to-files pipeline: what the copy engines and the copy kernels do per frame
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/io_copies; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/kt -o r -- python3 bench.py --io-only --io-simple --steps 40 > $OUT/line.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
kt = (glob.glob("$OUT/kt/*/*kernel_trace.csv") + glob.glob("$OUT/kt/*kernel_trace.csv"))[0]
mc = (glob.glob("$OUT/kt/*/*memory_copy_trace.csv") + glob.glob("$OUT/kt/*memory_copy_trace.csv"))
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(kt))]
marks = sorted(e for s_, e, n in rows if 'k_final_rows' in n)
a, b = marks[-40], marks[-10]
cb = [(e - s) / 1e3 for s, e, n in rows if 'copyBuffer' in n and s >= a and e <= b]
cb.sort()
print('copyBuffer kernels in the window: %d (%.1f per frame), total %.2f ms per frame; durations us: min %.1f median %.1f p90 %.1f max %.1f' % (len(cb), len(cb) / 30, sum(cb) / 30e3, cb[0], cb[len(cb) // 2], cb[int(len(cb) * 0.9)], cb[-1]))
big = [d for d in cb if d > 200]
print('  of them > 200 us: %d, total %.2f ms per frame' % (len(big), sum(big) / 30e3))
if mc:
    cop = list(csv.DictReader(open(mc[0])))
    print('memory-copy trace columns:', list(cop[0].keys()))
    acc = collections.defaultdict(lambda: [0, 0, 0.0])
    for r in cop:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s < a or e > b: continue
        nb = int(r.get('Size', r.get('Bytes', 0)) or 0)
        key = (r.get('Direction', r.get('Kind', '?')), 'big' if nb > (1 << 20) else 'small')
        acc[key][0] += 1; acc[key][1] += nb; acc[key][2] += (e - s) / 1e6
    for k, v in sorted(acc.items()):
        print('%-40s %6.1f copies/frame %8.1f MB/frame %7.2f ms/frame (sum of durations)' % (k, v[0] / 30, v[1] / 30e6, v[2] / 30))
PY
rm -rf $OUT/kt
cat $OUT/summary.txt
