"""Merge two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM traffic.

    python3 tools/pmc_traffic.py <dir FETCH_SIZE pass> <dir WRITE_SIZE pass> [commit] > traffic.json

The summary records the SHA-256 of the HIP sources the profiled kernels come from: bench.py reports `traffic` only while
those files are unchanged.

Counter unit: KB.  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE reports
half of a wide coalesced read (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import glob
import json
import re
import sys


def load(d, counter):
    f = (glob.glob(d + '/*counter_collection.csv') + glob.glob(d + '/*/*counter_collection.csv'))[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').strip()
        a = acc.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return acc


STAMPED = ['blackbox_amd/csrc/bbx_zogy3.hip', 'blackbox_amd/csrc/bbx_calibrate.hip', 'blackbox_amd/csrc/bbx_lacosmic.hip']


def main():
    import hashlib
    import os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    fe, wr = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    out = {'commit': sys.argv[3] if len(sys.argv) > 3 else 'unknown',
           'source_sha256': {f: hashlib.sha256(open(os.path.join(root, f), 'rb').read()).hexdigest() for f in STAMPED},
           '_note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 3 '
                    '--warmup 1 --no-cpu --no-extras --lanes 1 --depth 2` on MI355X; counter unit KB; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 '
                    '(gfx950: FETCH_SIZE reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section; '
                    '8-B and 4-B/lane loads are uncalibrated)',
           'kernels': {}}
    for k, (n, v) in fe.items():
        if not (k.startswith('k_') or k.startswith('z3::') or k.startswith('z2::') or 'rocclr' in k):
            continue
        w = wr.get(k, [n, 0.0])
        fk, wk = v / n, w[1] / max(1, w[0])
        rec = dict(launches=n, FETCH_SIZE_KB=fk, WRITE_SIZE_KB=wk, traffic_bytes_per_launch=(2 * fk + wk) * 1024)
        out['kernels'][k] = rec
        # the names bench.py uses: kernels of bbx_zogy_frame without namespace / plan, the calibration kernels without variant
        short = re.sub(r'<.*', '', k).split('::')[-1]
        short = {'k_calibrate_v4': 'k_calibrate', 'k_lac_cand_v4': 'k_lac_cand', 'k_img_rows_both': 'k_img_rows',
                 'k_spline_zoom4': 'k_spline_zoom'}.get(short, short)
        if short != k and short not in out['kernels']:
            out['kernels'][short] = dict(rec, alias_of=k)
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
