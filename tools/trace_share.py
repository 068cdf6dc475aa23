"""Attribution of the pipeline's wall time to kernels: in a kernel trace of the running pipeline, every instant of the steady
window is split evenly among the kernels running at it (1 / k each when k overlap); the shares are summed per kernel name
and divided by the frames of the window.  The column adds up to the frame time (plus idle); a kernel that always runs beside
others gets a fraction of its duration, one that runs alone all of it.
    python3 tools/trace_share.py <rocprofv3 output dir> [marker kernel = k_final_rows]"""
import csv, glob, sys, collections


def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else 'k_final_rows'
    f = (glob.glob(d + '/*/*kernel_trace.csv') + glob.glob(d + '/*kernel_trace.csv'))[0]
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
    marks = sorted(e for s, e, n in rows if marker in n)
    nfr = min(60, len(marks) - 20)
    a, b = marks[-nfr - 10], marks[-10]
    ev = []
    for i, (s, e, n) in enumerate(rows):
        if e <= a or s >= b:
            continue
        ev.append((max(s, a), 1, i)); ev.append((min(e, b), -1, i))
    ev.sort()
    active = set()
    share = collections.defaultdict(float); dur = collections.defaultdict(float); cnt = collections.defaultdict(int)
    idle = 0.0
    t_prev = a
    for t, kind, i in ev:
        dt = t - t_prev
        if dt > 0:
            if active:
                w = dt / len(active)
                for j in active:
                    share[rows[j][2]] += w
            else:
                idle += dt
        t_prev = t
        if kind == 1:
            active.add(i); cnt[rows[i][2]] += 1
        else:
            active.discard(i)
    for s, e, n in rows:
        if not (e <= a or s >= b):
            dur[n] += min(e, b) - max(s, a)
    short = lambda n: n.split('(')[0].replace('void ', '')[:58]
    print('window %.1f ms, %d frames: %.3f ms per frame; idle %.3f ms per frame' % ((b - a) / 1e6, nfr, (b - a) / 1e6 / nfr, idle / 1e6 / nfr))
    print('%-58s %9s %9s %7s' % ('kernel', 'share ms', 'dur ms', 'calls'))
    tot = 0
    for n, sh in sorted(share.items(), key=lambda x: -x[1])[:45]:
        print('%-58s %9.3f %9.3f %7.1f' % (short(n), sh / 1e6 / nfr, dur[n] / 1e6 / nfr, cnt[n] / nfr))
    print('sum of shares %.3f ms per frame' % (sum(share.values()) / 1e6 / nfr))


main()
