"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-frame kernel times."""
import csv, glob, sys
d, nframes = sys.argv[1], float(sys.argv[2])
f = (glob.glob(d + '/*/*_kernel_stats.csv') + glob.glob(d + '/*_kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows:
    n = r['Name']
    if 'k_' in n[:14] or 'rocclr' in n:
        per = float(r['TotalDurationNs']) / 1e3 / nframes
        tot += per
        if per > 3:
            print('%-58s calls/frame=%5.1f us/frame=%8.1f avg_us=%8.1f' % (n[:58], int(r['Calls']) / nframes, per, float(r['AverageNs']) / 1e3))
print('total bbx kernels per frame us', round(tot, 1))
