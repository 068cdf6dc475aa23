"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-frame kernel times.
usage: prof_summary.py <dir> <nframes> [min_us]
Kernels that only build the synthetic scene (torch random numbers, index_add of the star stamps, ...) run once per process,
not per frame: they are listed apart and left out of the per-frame total."""
import csv, glob, sys
d, nframes = sys.argv[1], float(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
f = (glob.glob(d + '/*/*_kernel_stats.csv') + glob.glob(d + '/*_kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
tot = 0
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
SETUP = ('distribution_', 'indexFunc', 'index_add', 'randn', 'normal_', 'philox', 'cumsum', 'sort', 'scatter', 'index_put', 'arange', 'gather', 'index_elementwise')
# (a kernel of the scene runs a few times per process; the library's own radix sort runs every frame)
setup = [r for r in rows if not r['Name'].startswith('k_') and 'z3::' not in r['Name'] and any(k in r['Name'] for k in SETUP) and int(r['Calls']) < nframes]
rows = [r for r in rows if r not in setup]
for r in rows:
    n = r['Name']
    per = float(r['TotalDurationNs']) / 1e3 / nframes
    tot += per
    if per > min_us:
        print('%-70s calls/frame=%6.1f us/frame=%9.1f avg_us=%9.1f' % (n[:70], int(r['Calls']) / nframes, per, float(r['AverageNs']) / 1e3))
print('total kernels per frame us', round(tot, 1))
if setup:
    print('# once per process (synthetic scene): %.1f us in %d launches of %s' % (
        sum(float(r['TotalDurationNs']) for r in setup) / 1e3, sum(int(r['Calls']) for r in setup),
        sorted({r['Name'][:40] for r in setup})[:6]))
