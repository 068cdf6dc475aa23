"""From a rocprofv3 --kernel-trace CSV: how busy the GPU was (union of kernel intervals) over the middle 60 % of the run, and the
time covered by kernels of the given name patterns.   usage: trace_busy.py <dir> [pattern ...]"""
import csv, glob, sys
d = sys.argv[1]
pats = sys.argv[2:] or ['z3::']
f = (glob.glob(d + '/*/*kernel_trace.csv') + glob.glob(d + '/*kernel_trace.csv'))[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# the window: inside the last pipeline run -- between the frames 40 and 10 before the end (by their k_final_rows / k_calibrate launch)
marks = [e for s_, e, n in rows if 'k_final_rows' in n] or [e for s_, e, n in rows if 'k_calibrate' in n]
a, b = (marks[-40], marks[-10]) if len(marks) >= 45 else (rows[0][0], max(r[1] for r in rows))
nframes = 30 if len(marks) >= 45 else 0
def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        s, e = max(s, a), min(e, b)
        if e <= s: continue
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot
span = b - a
print('window %.1f ms (%d frames: %.2f ms per frame), %d kernels in the trace' % (span / 1e6, nframes, span / 1e6 / max(1, nframes), len(rows)))
print('any kernel running: %.1f %%' % (100 * union([(s, e) for s, e, n in rows]) / span))
for p in pats:
    print('%-12s running: %.1f %%' % (p, 100 * union([(s, e) for s, e, n in rows if p in n]) / span))
big = ('z3::', 'k_calibrate', 'k_xtalk', 'k_bkg_boxstats', 'k_spline_zoom', 'k_canny', 'k_compact_abs', 'k_bin2', 'k_lac_cand_v4', 'k_bsel_feed_v4', 'k_ccf_acc', 'k_hough')
print('GPU-filling kernels running: %.1f %%' % (100 * union([(s, e) for s, e, n in rows if any(q in n for q in big)]) / span))
# sum of durations of the GPU-filling kernels / window (> 100 % means they overlap each other)
print('sum of their durations: %.1f %% of the window' % (100 * sum(min(e, b) - max(s, a) for s, e, n in rows if any(q in n for q in big) and min(e, b) > max(s, a)) / span))
