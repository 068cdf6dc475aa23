"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-frame kernel times.
usage: prof_summary.py <dir> <nframes> [min_us]"""
import csv, glob, sys
d, nframes = sys.argv[1], float(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
f = (glob.glob(d + '/*/*_kernel_stats.csv') + glob.glob(d + '/*_kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
tot = 0
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows:
    n = r['Name']
    per = float(r['TotalDurationNs']) / 1e3 / nframes
    tot += per
    if per > min_us:
        print('%-70s calls/frame=%6.1f us/frame=%9.1f avg_us=%9.1f' % (n[:70], int(r['Calls']) / nframes, per, float(r['AverageNs']) / 1e3))
print('total kernels per frame us', round(tot, 1))
