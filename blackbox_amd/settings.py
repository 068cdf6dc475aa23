"""Reduction parameters for the hot path, under the reference's own names.

Mirror of the rows of ``Settings/set_blackbox.py`` that the per-image reduction
reads (file:line given per entry) plus the ``set_zogy`` values the reference
takes from the (external) ZOGY settings module.  Values that are dictionaries
are keyed by telescope ('ML1', 'BG2', ...) or by telescope family ('BG') and
are resolved with :func:`get_par`, like the reference's ``get_par(par, tel)``.
"""

# ---- reduction switches (set_blackbox.py:36-52) --------------------------------
subtract_mbias = {'ML1': False, 'BG': True}
detect_sats = True
correct_nonlin = False
voscan_poldeg = 3
ncal_max = {'bias': 20, 'dark': 20, 'flat': 15}

# ---- LA-Cosmic (set_blackbox.py:211-218) ---------------------------------------
sigclip = {'ML1': 15, 'BG': 20}
sigfrac = 0.01
objlim = 3
niter = 3
sepmed = False

# ---- satellite trails (set_blackbox.py:222-228) --------------------------------
use_asta = False          # ASTA's Keras model cannot be shipped; classical path
sat_bin = 2               # binning of the classical (acstools-like) path, blackbox.py:4163

# ---- CCD (set_blackbox.py:241-337) ----------------------------------------------
gain = {
    'ML1': [2.112, 2.125, 2.130, 2.137, 2.156, 2.158, 2.163, 2.164,
            2.109, 2.124, 2.126, 2.132, 2.136, 2.154, 2.155, 2.157],
    'BG2': [2.694, 2.685, 2.691, 2.661, 2.655, 2.673, 2.695, 2.659,
            2.654, 2.748, 2.712, 2.717, 2.714, 2.702, 2.673, 2.743],
    'BG3': [2.614, 2.609, 2.634, 2.647, 2.600, 2.616, 2.683, 2.649,
            2.680, 2.679, 2.644, 2.604, 2.615, 2.633, 2.615, 2.714],
    'BG4': [2.415, 2.393, 2.365, 2.333, 2.340, 2.320, 2.348, 2.389,
            2.395, 2.403, 2.381, 2.350, 2.362, 2.369, 2.391, 2.430],
}
satlevel = {
    'ML1': [5.89e4, 5.94e4, 5.82e4, 5.59e4, 5.60e4, 5.63e4, 5.60e4, 5.75e4,
            5.88e4, 5.81e4, 5.71e4, 5.65e4, 5.59e4, 5.60e4, 5.59e4, 5.65e4],
    'BG2': [3.84e4, 3.77e4, 3.75e4, 3.79e4, 3.79e4, 3.80e4, 3.75e4, 3.93e4,
            4.50e4, 4.08e4, 4.08e4, 4.09e4, 4.07e4, 3.95e4, 4.15e4, 4.37e4],
    'BG3': [3.96e4, 3.83e4, 3.79e4, 3.77e4, 3.81e4, 3.83e4, 3.74e4, 3.94e4,
            4.00e4, 3.98e4, 4.13e4, 4.29e4, 4.29e4, 4.22e4, 4.13e4, 4.38e4],
    'BG4': [4.11e4, 4.09e4, 4.16e4, 4.29e4, 4.32e4, 4.29e4, 4.23e4, 4.41e4,
            4.66e4, 4.60e4, 4.53e4, 4.67e4, 4.66e4, 4.65e4, 4.64e4, 4.66e4],
}
flat_norm_sec = {'ML1': (slice(6600, 9240), slice(5280, 7920)),
                 'BG2': (slice(500, 2000), slice(1320, 6600)),
                 'BG3': (slice(300, 1200), slice(5280, 10000)),
                 'BG4': (slice(2640, 5280), slice(3960, 7920))}
ny, nx = 2, 8
ysize_chan, xsize_chan = 5280, 1320

# rows of the data section searched for saturated columns by os_corr
# (blackbox.py:6625)
os_ypix_lim = {'BG2': (2640, 5280), 'BG3': (1320, 2640), 'BG4': (1320, 2640)}

# ---- set_zogy values used by the reduction (external module upstream) -----------
mask_value = {'bad': 1, 'cosmic ray': 2, 'saturated': 4,
              'saturated-connected': 8, 'satellite trail': 16, 'edge': 32,
              'crosstalk': 64}
bkg_boxsize = 60
bkg_filtersize = 3
subimage_size = 1320
subimage_border = 40
transient_nsigma = 6

# calibration files (explicit paths; the date-based master selection of
# master_prep, blackbox.py:4625-4905, is orchestration and out of scope)
bad_pixel_mask = None      # path containing 'bpm' -> 'bpm_{filt}' (blackbox.py:4386)
crosstalk_file = None
master_flat = None
master_bias = None


def get_par(par, tel):
    """value of [par] for telescope [tel]: exact key, then the alphabetic
    prefix ('BG2' -> 'BG'), else the parameter itself (zogy.get_par)."""
    if isinstance(par, dict):
        if tel in par:
            return par[tel]
        base = ''.join(c for c in str(tel) if c.isalpha())
        if base in par:
            return par[base]
    return par
