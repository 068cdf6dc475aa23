"""TEST INFRASTRUCTURE -- runs ONLY in the build container, never on the GPU box.

Imports the reference's own ``blackbox.py`` (read-only at /root/reference) under
the container's /opt/conda/bin/python3.9 so that its own functions
(define_sections, gain_corr, os_corr, mask_init, fill_sat_holes, xtalk_corr,
mask_header) can be executed to produce golden vectors (oracle/gen_golden.py).

Third-party modules that blackbox.py imports at module level but that are not
installed here (set_zogy, zogy, match2SSO, astroscrappy, acstools, ephem,
fitsio, ASTA) are replaced by empty placeholder modules; none of the functions
we call touches them, except for a handful of tiny zogy helpers
(get_par, read_hdulist, isfile, ...) which are restated below -- those helpers
are OUR code, not the reference's (SURVEY.md section 8c spells out the caveat).
Nothing from /root/reference is copied into the repository.
"""
import os
import sys
import types

os.environ.setdefault('HOME', '/tmp')          # set_blackbox.py:71
import numpy as np

# conda's astropy 4.3.1 / matplotlib 3.4.3 predate its numpy 1.26
for _n, _f in {'asscalar': lambda a: a.item(), 'alen': len}.items():
    if not hasattr(np, _n):
        setattr(np, _n, _f)
import warnings
warnings.filterwarnings('ignore', message='A NumPy version')
from astropy.utils import iers
iers.conf.auto_download = False                 # blackbox.py:114-119, no network
import matplotlib
import matplotlib.cm
if not hasattr(matplotlib, 'colormaps'):
    matplotlib.colormaps = matplotlib.cm

REF = '/root/reference'
sys.path[:0] = [REF, REF + '/Settings']

MASK_VALUE = {'bad': 1, 'cosmic ray': 2, 'saturated': 4,
              'saturated-connected': 8, 'satellite trail': 16, 'edge': 32,
              'crosstalk': 64}


def _stub(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def load(cal_dir='/tmp/bbx_cal'):
    """returns (blackbox module, set_blackbox module)"""
    if 'blackbox' in sys.modules:
        return sys.modules['blackbox'], sys.modules['set_blackbox']
    _stub('set_zogy', timing=False, display=False, make_plots=False,
          cal_dir=cal_dir, subimage_size=1320, mask_value=MASK_VALUE,
          obs_lat=-32.38, obs_lon=20.81, obs_height=1802,
          obs_timezone='Africa/Johannesburg')
    for n in ['set_match2SSO', 'match2SSO', 'astroscrappy', 'ephem', 'fitsio',
              'acstools']:
        _stub(n)
    _stub('acstools.satdet', detsat=None, make_mask=None)
    _stub('ASTA', ASTA=None)

    import astropy.io.fits as fits
    from astropy.table import Table, vstack, unique
    from astropy.stats import sigma_clip
    from scipy import ndimage, interpolate
    import subprocess
    import traceback
    import argparse

    def get_par(par, tel):                      # restatement of zogy.get_par
        if isinstance(par, dict):
            if tel in par:
                return par[tel]
            base = ''.join(c for c in str(tel) if c.isalpha())
            if base in par:
                return par[base]
        return par

    def get_rand_indices(shape, fraction=0.2):  # restatement; unseeded upstream
        n = int(np.prod(shape) * fraction)
        return tuple(np.random.randint(shape[i], size=n) for i in range(len(shape)))

    def read_hdulist(fits_file, get_data=True, get_header=False,
                     ext_name_indices=None, dtype=None, memmap=True):
        # minimal restatement: last HDU, optional dtype cast
        with fits.open(fits_file, memmap=False) as hdulist:
            data = hdulist[-1].data
            header = hdulist[-1].header
            if dtype is not None:
                data = data.astype(dtype, copy=False)
        if get_data and get_header:
            return data, header
        return data if get_data else header

    def list_files(path, search_str='', end_str='', start_str=None, recursive=False):
        # restatement of the zogy helper: files whose path starts with [path], contains
        # [search_str] and ends with [end_str]
        import glob
        names = glob.glob(path + '*') + (glob.glob(path + '/**/*', recursive=True) if recursive else [])
        return sorted(f for f in set(names) if os.path.isfile(f) and search_str in f and f.endswith(end_str)
                      and (start_str is None or os.path.basename(f).startswith(start_str)))

    _stub('zogy', np=np, fits=fits, list_files=list_files, Table=Table, vstack=vstack, unique=unique,
          sigma_clip=sigma_clip, ndimage=ndimage, interpolate=interpolate,
          subprocess=subprocess, traceback=traceback, argparse=argparse,
          get_par=get_par, get_rand_indices=get_rand_indices,
          read_hdulist=read_hdulist,
          isfile=os.path.isfile, isdir=os.path.isdir,
          mem_use=lambda *a, **k: None,
          log_timing_memory=lambda *a, **k: None, format_cat=None)

    import blackbox as bb
    import set_blackbox as set_bb
    return bb, set_bb
