"""Catalogue products of the subtraction stage: the binary
tables `_red_cat.fits` / `_red_trans.fits` (set_blackbox.py:160-164) and their empty "dummy"
versions for red-flagged images (qc.py:451-503 -> zogy.format_cat).

[EXT] zogy.format_cat is not part of /root/reference; the column names follow the BlackGEM /
MeerLICHT catalogue conventions for the quantities the hot path produces (pixel positions,
PSF-weighted optimal fluxes, ZOGY significance and PSF flux); everything that needs astrometry,
photometric calibration or the real-bogus classifier (RA/DEC, MAG_*, CLASS_REAL, thumbnails) is
out of scope and absent.
"""
import numpy as np

from . import fitsio

COLUMNS = {
    'new': (('NUMBER', np.int32, ''), ('X_POS', np.float32, 'pix'), ('Y_POS', np.float32, 'pix'),
            ('E_FLUX_PEAK', np.float32, 'e-'), ('E_FLUX_OPT', np.float32, 'e-'), ('E_FLUXERR_OPT', np.float32, 'e-'),
            ('SNR_OPT', np.float32, '')),
    'trans': (('NUMBER', np.int32, ''), ('X_PEAK', np.int32, 'pix'), ('Y_PEAK', np.int32, 'pix'),
              ('SNR_ZOGY', np.float32, ''), ('E_FLUX_ZOGY', np.float32, 'e-'), ('E_FLUXERR_ZOGY', np.float32, 'e-')),
}
COLUMNS['ref'] = COLUMNS['new']


def format_cat(table, cat_output, cat_type='new', header2add=None):
    """zogy.format_cat(cat_in, cat_out, cat_type=, header2add=): write the catalogue of type
    [cat_type] with its fixed column set; table None -> zero rows (dummy catalogue)"""
    if cat_type not in COLUMNS:
        raise ValueError('cat_type {} not in {}'.format(cat_type, sorted(COLUMNS)))
    n = 0 if table is None else len(next(iter(table.values()))) if table else 0
    cols, units = {}, {}
    for name, dt, unit in COLUMNS[cat_type]:
        if table is not None and name in table:
            cols[name] = np.asarray(table[name]).astype(dt)
        elif name == 'NUMBER':
            cols[name] = np.arange(1, n + 1, dtype=dt)
        else:
            cols[name] = np.zeros(n, dtype=dt)
        units[name] = unit
    fitsio.write_table(cat_output, cols, header2add, units=units)
    return cat_output


def transient_table(transients):
    """list of dict(y, x, scorr, fpsf, fpsferr) (zogy.optimal_subtraction) -> column dict (FITS
    pixel coordinates, 1-based)"""
    t = transients or []
    return dict(X_PEAK=np.array([d['x'] + 1 for d in t], np.int32), Y_PEAK=np.array([d['y'] + 1 for d in t], np.int32),
                SNR_ZOGY=np.array([d['scorr'] for d in t], np.float32),
                E_FLUX_ZOGY=np.array([d['fpsf'] for d in t], np.float32),
                E_FLUXERR_ZOGY=np.array([d['fpsferr'] for d in t], np.float32))


def write_small_products(jobs):
    """the small files of a frame in one call (a worker process of the host pool runs it for blackbox.py's list run, so
    that their formatting does not hold the interpreter lock of the process that drives the GPU): jobs = [(kind, args)],
    kind in 'image' (fitsio.write_image), 'header' (fitsio.write_header), 'cat' (format_cat), 'trans' (format_cat of
    transient_table(args[0]))"""
    done = []
    for kind, args in jobs:
        if kind == 'image':
            fitsio.write_image(*args)
        elif kind == 'header':
            fitsio.write_header(*args)
        elif kind == 'cat':
            format_cat(args[0], args[1], cat_type=args[2], header2add=args[3])
        elif kind == 'trans':
            format_cat(transient_table(args[0]), args[1], cat_type='trans', header2add=args[2])
        else:
            raise ValueError('unknown small product {!r}'.format(kind))
        done.append(args[1] if kind in ('cat', 'trans') else args[0])
    return done
