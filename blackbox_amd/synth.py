"""Seeded synthetic MeerLICHT/BlackGEM CCD frames (numpy only, host side).

Used by the golden-vector generator (oracle/gen_golden.py, runs under the
container's python3.9 / numpy 1.26), by the tests and by bench.py (python3.10 /
numpy 2.2).  To make the very same pixels come out under both interpreters the
generator restricts itself to operations that are bit-reproducible everywhere:

* random numbers only from ``numpy.random.RandomState.random_sample`` (the
  legacy MT19937 stream is frozen by numpy policy),
* IEEE-exact arithmetic only: + - * / sqrt floor.  No exp/log/pow/sin/cos.
  Gaussian-like noise is an Irwin-Hall sum of 4 uniforms, stars are Moffat
  beta=2 profiles 1/(1+r^2/a^2)^2, the bias "exponential" rise in the
  horizontal overscan is a rational function.

The geometry follows the reference's define_sections (blackbox.py:6334-6402):
ny x nx = 2 x 8 channels, each channel = data section (ysize_chan x xsize_chan)
plus a vertical overscan strip to its right and horizontal overscan rows
between the two channel rows.
"""
import numpy as np

NY, NX = 2, 8          # set_blackbox.py:335
SQRT3 = 3.0 ** 0.5

from .settings import gain as GAIN      # set_blackbox.py:241-281


def _gauss(rs, shape):
    """unit-variance, zero-mean Irwin-Hall(4) deviates, exact arithmetic"""
    u = rs.random_sample((4,) + tuple(shape))
    return (u[0] + u[1] + u[2] + u[3] - 2.0) * SQRT3


def raw_shape(ysize_chan, xsize_chan, os_y=20, os_x=180):
    return (NY * (ysize_chan + os_y), NX * (xsize_chan + os_x))


def make_scene(ysize_chan, xsize_chan, seed, n_stars=40, n_sat=2, n_cr=30,
               sky=250.0, trail=False):
    """Noise-free scene in electrons on the reduced (NY*ysize, NX*xsize) grid,
    plus the cosmic-ray layer (kept apart so tests know the truth)."""
    rs = np.random.RandomState(seed)
    ny, nx = NY * ysize_chan, NX * xsize_chan
    scene = np.zeros((ny, nx)) + sky
    yy, xx = np.mgrid[0:ny, 0:nx]
    # slow sky gradient (exact: linear)
    scene += 0.05 * sky * (xx / float(nx) - 0.5) + 0.03 * sky * (yy / float(ny) - 0.5)

    def add_star(x0, y0, flux, fwhm):
        a = fwhm / (2.0 * (2.0 ** 0.5 - 1.0) ** 0.5)      # Moffat beta=2
        r = int(4 * fwhm) + 2
        xl, xh = max(0, int(x0) - r), min(nx, int(x0) + r + 1)
        yl, yh = max(0, int(y0) - r), min(ny, int(y0) + r + 1)
        dx = xx[yl:yh, xl:xh] - x0
        dy = yy[yl:yh, xl:xh] - y0
        q = 1.0 + (dx * dx + dy * dy) / (a * a)
        scene[yl:yh, xl:xh] += flux / (3.141592653589793 * a * a) / (q * q)

    u = rs.random_sample((n_stars, 4))
    for k in range(n_stars):
        # power-law-ish fluxes 1e3..1e6 e- via exact ops: 1e3 * (1 + 999 u^4)
        flux = 1.0e3 * (1.0 + 999.0 * u[k, 2] * u[k, 2] * u[k, 2] * u[k, 2])
        add_star(u[k, 0] * nx, u[k, 1] * ny, flux, 3.0 + 2.0 * u[k, 3])
    u = rs.random_sample((max(n_sat, 1), 3))
    for k in range(n_sat):
        add_star(20 + u[k, 0] * (nx - 40), 8 + u[k, 1] * (ny - 16),
                 2.0e7 * (1.0 + u[k, 2]), 4.0)
    if trail:
        # one straight trail, width ~6 px (Moffat-like cross section), 150 e-/px
        x0, x1 = 0.0, float(nx)
        y0, y1 = 0.2 * ny, 0.7 * ny
        norm = ((x1 - x0) ** 2 + (y1 - y0) ** 2) ** 0.5
        d = ((xx - x0) * (y1 - y0) - (yy - y0) * (x1 - x0)) / norm
        q = 1.0 + d * d / 9.0
        scene += 150.0 / (q * q)

    cr = np.zeros((ny, nx))
    u = rs.random_sample((max(n_cr, 1), 6))
    for k in range(n_cr):
        x0 = int(3 + u[k, 0] * (nx - 6))
        y0 = int(3 + u[k, 1] * (ny - 6))
        length = 1 + int(u[k, 2] * 12)
        amp = 300.0 + 2.0e4 * u[k, 3] * u[k, 3]
        # direction on a coarse grid: steps in {-1,0,1}
        sx = int(u[k, 4] * 3) - 1
        sy = int(u[k, 5] * 3) - 1
        if sx == 0 and sy == 0:
            sx = 1
        for j in range(length):
            x, y = x0 + j * sx, y0 + j * sy
            if 0 <= x < nx and 0 <= y < ny:
                cr[y, x] += amp
    return scene, cr


def make_flat(ysize_chan, xsize_chan, seed):
    rs = np.random.RandomState(seed + 7001)
    ny, nx = NY * ysize_chan, NX * xsize_chan
    yy, xx = np.mgrid[0:ny, 0:nx]
    r2 = ((xx - 0.5 * nx) / (0.5 * nx)) ** 2 + ((yy - 0.5 * ny) / (0.5 * ny)) ** 2
    flat = 1.0 - 0.05 * r2 + 0.005 * _gauss(rs, (ny, nx))
    return flat.astype(np.float32)


def make_bias(ysize_chan, xsize_chan, seed):
    rs = np.random.RandomState(seed + 7002)
    ny, nx = NY * ysize_chan, NX * xsize_chan
    return _gauss(rs, (ny, nx)).astype(np.float32)


def make_bpm(ysize_chan, xsize_chan, seed, edge=6, frac_bad=2e-4):
    """uint8 bad-pixel mask: bad=1, edge=32 (set_zogy.mask_value)"""
    rs = np.random.RandomState(seed + 7003)
    ny, nx = NY * ysize_chan, NX * xsize_chan
    bpm = np.zeros((ny, nx), dtype=np.uint8)
    bpm[rs.random_sample((ny, nx)) < frac_bad] = 1
    if edge > 0:
        bpm[:edge, :] = 32
        bpm[-edge:, :] = 32
        bpm[:, :edge] = 32
        bpm[:, -edge:] = 32
    return bpm


def make_xtalk(seed, scale=2e-4):
    """(victim, source, correction) rows, 1-based channels, 240 coefficients
    (format of blackbox.py:7157-7161)"""
    rs = np.random.RandomState(seed + 7004)
    u = rs.random_sample((16, 16))
    rows = []
    for v in range(16):
        for s in range(16):
            if v != s:
                rows.append((v + 1, s + 1, scale * u[v, s]))
    return rows


def write_xtalk(path, rows):
    with open(path, 'w') as f:
        f.write('victim source correction\n')
        for v, s, c in rows:
            f.write('{} {} {!r}\n'.format(v, s, float(c)))


def make_raw(ysize_chan, xsize_chan, seed, tel='ML1', os_y=20, os_x=180,
             flat=None, bias=None, scene=None, cr=None, bias_adu=3000.0,
             rdnoise_adu=4.0, hos_bleed=False):
    """Raw uint16 frame with overscans.  Inverse of the calibration: a scene in
    e- goes through x flat, + master bias, / gain, + per-channel bias level with
    a cubic row trend and the horizontal-overscan column structure, + read
    noise, rounded to ADU and clipped to uint16."""
    rs = np.random.RandomState(seed + 7005)
    gain = GAIN[tel]
    dy, dx = ysize_chan + os_y, xsize_chan + os_x
    raw = np.zeros((NY * dy, NX * dx))
    if scene is None:
        scene, cr = make_scene(ysize_chan, xsize_chan, seed)
    img = scene + (cr if cr is not None else 0.0)
    # shot noise (Gaussian approximation, exact ops)
    img = img + np.sqrt(np.maximum(img, 0.0)) * _gauss(rs, img.shape)
    if flat is not None:
        img = img * flat.astype(np.float64)
    if bias is not None:
        img = img + bias.astype(np.float64)
    ub = rs.random_sample((16, 4))
    yrow = np.arange(dy, dtype=np.float64)
    xcol = np.arange(dx, dtype=np.float64)
    for c in range(16):
        iy, ix = c // NX, c % NX
        y0, x0 = iy * dy, ix * dx
        level = bias_adu + 150.0 * (ub[c, 0] - 0.5)
        t = yrow / dy - 0.5
        trend = 2.0 * ub[c, 1] * t + 3.0 * (ub[c, 2] - 0.5) * t * t * t
        q = 1.0 + xcol / 20.0
        colstruct = 20.0 * (0.5 + ub[c, 3]) / (q * q)
        chan = np.zeros((dy, dx)) + level + trend[:, None]
        chan = chan + rdnoise_adu * (1.0 + 0.3 * (ub[c, 3] - 0.5)) * _gauss(rs, (dy, dx))
        # data section rows inside the channel: lower row [0, ysize), upper row
        # [os_y, dy)  (define_sections: data_sec y origin = dy + ysize_os)
        ys = 0 if iy == 0 else os_y
        sec = img[iy * ysize_chan:(iy + 1) * ysize_chan,
                  ix * xsize_chan:(ix + 1) * xsize_chan]
        chan[ys:ys + ysize_chan, :xsize_chan] += sec / gain[c]
        # column structure of the bias shows up in the data columns and in the
        # horizontal overscan rows alike (that is what os_corr fits)
        chan[:, :xsize_chan] += colstruct[None, :xsize_chan]
        if hos_bleed:
            # charge bleeding into the horizontal overscan rows: a 3-column
            # block (masked + dilated by os_corr, blackbox.py:6590-6614) and one
            # isolated column (restored by the binary_opening trick)
            yo = ysize_chan if iy == 0 else 0
            cb = 40 + int(ub[c, 0] * (xsize_chan - 80))
            chan[yo:yo + os_y, cb:cb + 3] += 3000.0
            ci = 40 + int(ub[c, 1] * (xsize_chan - 80))
            chan[yo:yo + os_y, ci] += 2500.0
        raw[y0:y0 + dy, x0:x0 + dx] = chan
    raw = np.floor(raw + 0.5)
    raw = np.minimum(np.maximum(raw, 0.0), 65535.0)
    return raw.astype(np.uint16)


def make_case(ysize_chan, xsize_chan, seed, tel='ML1', os_y=20, os_x=45,
              with_bias=False, hos_bleed=False, **scene_kw):
    """Everything one reduction needs, keyed like the golden fixtures."""
    scene, cr = make_scene(ysize_chan, xsize_chan, seed, **scene_kw)
    flat = make_flat(ysize_chan, xsize_chan, seed)
    bias = make_bias(ysize_chan, xsize_chan, seed) if with_bias else None
    raw = make_raw(ysize_chan, xsize_chan, seed, tel=tel, os_y=os_y, os_x=os_x,
                   flat=flat, bias=bias, scene=scene, cr=cr, hos_bleed=hos_bleed)
    return dict(raw=raw, flat=flat, bias=bias,
                bpm=make_bpm(ysize_chan, xsize_chan, seed),
                xtalk=make_xtalk(seed), scene=scene, cr=cr)


# ---- inputs of the round-2 reference-run fixtures (oracle/gen_golden_r02.py, tests) -----------
# master frames: channels wider than 200 px (master_prep's boundary strips are 200 columns)
MASTER_GEOM = (64, 210)
MASTER_NORM_SEC = (slice(32, 96), slice(400, 800))


def master_bpm():
    ys, xs = MASTER_GEOM
    bpm = np.zeros((2 * ys, 8 * xs), np.uint8)
    bpm[:3] = 32; bpm[-3:] = 32; bpm[:, :3] = 32; bpm[:, -3:] = 32
    bpm[40, 500] = 1
    return bpm


def master_frames(imgtype, n=6, seed=5):
    """-> (list of float32 frames, MEDSEC per frame or None): reduced flats with a level step per
    channel and a few non-positive pixels, or bias frames with zeros (sigma_clipped_stats(mask_value=0))"""
    ys, xs = MASTER_GEOM
    ny, nx = 2 * ys, 8 * xs
    rs = np.random.RandomState(seed + (0 if imgtype == 'flat' else 100))
    chan = (np.arange(nx) // xs)[None, :] + 8 * (np.arange(ny) // ys)[:, None]
    frames, medsec = [], []
    for k in range(n):
        if imgtype == 'flat':
            lev = 20000.0 + 1500.0 * k
            img = lev * (1.0 + 0.02 * _gauss(rs, (ny, nx))) * (1.0 + 0.004 * chan)
            img[10, 20 + k] = -5.0
            img = img.astype(np.float32)
            medsec.append(np.median(img[MASTER_NORM_SEC]))
        else:
            img = (3.0 * _gauss(rs, (ny, nx)) + 0.2 * chan).astype(np.float32)
            img[rs.random_sample((ny, nx)) < 0.001] = 0.0
        frames.append(img)
    return frames, (medsec if imgtype == 'flat' else None)


# get_flatstats: a square frame (the reference reshapes into nsubs x nsubs sub-images)
FLATSTAT_GEOM = (120, 30)
FLATSTAT_SUB = 60
FLATSTAT_SEC = (slice(60, 180), slice(30, 150))


def flatstat_frame(seed=9):
    ys, xs = FLATSTAT_GEOM
    ny, nx = 2 * ys, 8 * xs
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:ny, 0:nx]
    data = (30000.0 * (1.0 - 0.1 * ((yy - 120.0) ** 2 + (xx - 120.0) ** 2) / 28800.0) * (1.0 + 0.01 * _gauss(rs, (ny, nx)))).astype(np.float32)
    mask = np.zeros((ny, nx), np.uint8)
    mask[:4] = 32; mask[-4:] = 32; mask[:, :4] = 32; mask[:, -4:] = 32
    mask[rs.random_sample((ny, nx)) < 0.01] |= 1
    data[100, 100] = 250000.0
    return data, mask


NONLIN_GEOM = (32, 40)


def nonlin_frame(seed=13):
    """overscan-corrected frame in e-: counts from ~0 to above the 50000-count limit of nonlin_corr"""
    ys, xs = NONLIN_GEOM
    rs = np.random.RandomState(seed)
    return (rs.random_sample((2 * ys, 8 * xs)) * 130000.0).astype(np.float32)


def nonlin_splines():
    """inputs of 16 scipy UnivariateSpline fits (x, y, w, k, s): fractional non-linearity vs counts"""
    out = []
    for c in range(16):
        x = np.arange(0.0, 66000.0, 3000.0)
        y = 1e-3 * (1.0 + 0.1 * c) * ((x / 30000.0) * (x / 30000.0) - x / 30000.0)
        out.append((x, y, None, 3, 0.0))
    return out


def qc_headers():
    """header dicts for the qc_check pins: (name, telescope, check_key_type, header)"""
    base = {'IMAGETYP': 'object', 'FILTER': 'q', 'RA': 120.0, 'DEC': -30.0, 'XTALK-P': True, 'NONLIN-P': False, 'GAIN-P': True, 'OS-P': True, 'MBIAS-P': False,
            'MFLAT-P': True, 'COSMIC-P': True, 'SAT-P': True, 'BIASMEAN': 6460.0, 'RDNOISE': 9.0, 'N-INFNAN': 0, 'NCOSMICS': 10.0,
            'NSATS': 1}
    cases = [('all_green', 'ML1', 'full', dict(base))]
    h = dict(base); h['XTALK-P'] = False; h['RDNOISE'] = 12.0
    cases.append(('red_flag_and_yellow', 'ML1', 'full', h))
    h = dict(base); h['BIASMEAN'] = 6900.0; h['NCOSMICS'] = 80.0; h['N-INFNAN'] = 5
    cases.append(('orange_sigma', 'ML1', 'full', h))
    h = dict(base); h['NCOSMICS'] = 'None'; h['COSMIC-P'] = False; h['NSATS'] = 150
    cases.append(('none_value_and_red', 'ML1', 'full', h))
    h = dict(base); h['MBIAS-P'] = True; h['BIASMEAN'] = 3300.0; h['RDNOISE'] = 16.0
    cases.append(('blackgem', 'BG3', 'full', h))
    return cases


CLIP_FILTERS = {'default': ((5, 1), (3.5, 4.0), (4, 1)), 'wide': ((7, 3, 1), (3.0, 3.5, 5.0), (6, 3, 1))}


def clip_points(seed=21):
    """clipped-pixel table of one input image for the pass_filters pin: isolated points, clusters
    (cosmic-ray / satellite residuals), both signs, points at the frame corners -> (x, y 1-based, nsigma, shape)"""
    rs = np.random.RandomState(seed)
    ny, nx = 120, 160
    xs, ys, ns = [], [], []
    for _ in range(150):
        xs.append(1 + int(rs.random_sample() * nx)); ys.append(1 + int(rs.random_sample() * ny)); ns.append(3.0 + 4.0 * rs.random_sample())
    for _ in range(8):
        cx, cy = 5 + int(rs.random_sample() * (nx - 10)), 5 + int(rs.random_sample() * (ny - 10))
        sign = 1.0 if rs.random_sample() < 0.7 else -1.0
        for _ in range(3 + int(rs.random_sample() * 12)):
            xs.append(cx + int(rs.random_sample() * 5)); ys.append(cy + int(rs.random_sample() * 5)); ns.append(sign * (3.2 + 6.0 * rs.random_sample()))
    for (x, y) in ((1, 1), (nx, ny), (nx, 1), (1, ny), (nx - 1, ny - 1), (nx, ny - 2), (nx - 2, ny)):
        xs.append(x); ys.append(y); ns.append(6.0)
    x, y, n = np.array(xs), np.array(ys), np.array(ns, np.float32)
    key = y * 100000 + x                                        # one entry per pixel
    _, first = np.unique(key, return_index=True)
    first.sort()
    return x[first], y[first], n[first], (ny, nx)


# ---- a12: binned frames for the satellite-trail front end (skimage fixtures, oracle, HIP) ---------
SAT_SCENES = {'trail': dict(seed=1, ny=256, nx=320, trail=(0.5, 200.0, 150.0, 1.5)),
              'faint': dict(seed=2, ny=300, nx=280, trail=(2.3, -60.0, 60.0, 2.0)),
              'none': dict(seed=3, ny=200, nx=360, trail=None)}


def sat_scene(name):
    """a 2x2-binned frame (float32, e-): sky + noise, 25 Gaussian stars of 10^3..10^5.5 e-, optionally a
    trail (theta [rad], rho [px], amplitude [e-], sigma [px]) x cos(theta) + y sin(theta) = rho"""
    p = SAT_SCENES[name]
    rs = np.random.RandomState(p['seed'])
    ny, nx = p['ny'], p['nx']
    b = (1000 + 30 * rs.standard_normal((ny, nx))).astype(np.float32)
    yy, xx = np.mgrid[0:ny, 0:nx]
    for _ in range(25):
        cy, cx, fl = rs.uniform(0, ny), rs.uniform(0, nx), 10 ** rs.uniform(3, 5.5)
        b += (fl / (2 * np.pi * 4) * np.exp(-0.5 * ((yy - cy) ** 2 + (xx - cx) ** 2) / 4)).astype(np.float32)
    if p['trail'] is not None:
        th, rho, amp, sig = p['trail']
        d = xx * np.cos(th) + yy * np.sin(th) - rho
        b += (amp * np.exp(-0.5 * (d / sig) ** 2)).astype(np.float32)
    return b


# ---- full-resolution scenes for the satellite-trail mask (oracle/gen_golden_sat.py -> tests/golden/sat_mask.npz) ----------
# (xa, ya, xb, yb, amplitude [e-], FWHM [px]) of a straight trail through (xa, ya), (xb, yb); frames of ny x nx pixels
SAT_MASK_SCENES = {'shallow': dict(seed=1, ny=600, nx=900, trail=(0, 120, 900, 470, 120.0, 6.0)),
                   'steep': dict(seed=2, ny=600, nx=900, trail=(200, 0, 520, 600, 80.0, 6.0)),
                   'level': dict(seed=4, ny=600, nx=900, trail=(0, 300, 900, 305, 200.0, 8.0)),
                   'falling': dict(seed=5, ny=700, nx=1100, trail=(0, 610, 1100, 40, 150.0, 5.0)),
                   'upright': dict(seed=6, ny=1000, nx=560, trail=(300, 0, 250, 1000, 160.0, 7.0))}


def sat_full_scene(seed, ny=600, nx=900, trail=None, nstars=150):
    """a reduced frame (float32, e-): sky 250 + noise 18, [nstars] Gaussian stars (sigma 1.7 px, 10^3..10^5.5 e-), a trail
    -> (frame, truth mask of the trail within its FWHM)"""
    rs = np.random.RandomState(seed)
    img = 250 + rs.normal(0, 18, (ny, nx))
    yy, xx = np.mgrid[0:ny, 0:nx]
    for _ in range(nstars):
        y0, x0, f = rs.uniform(0, ny), rs.uniform(0, nx), 10 ** rs.uniform(3, 5.5)
        r2 = (yy - y0) ** 2 + (xx - x0) ** 2
        sel = r2 < 15 ** 2
        img[sel] += (f / (2 * np.pi * 1.7 ** 2)) * np.exp(-r2[sel] / (2 * 1.7 ** 2))
    truth = np.zeros((ny, nx), bool)
    if trail is not None:
        (xa, ya, xb, yb, amp, width) = trail
        norm = np.hypot(xb - xa, yb - ya)
        d = ((xx - xa) * (yb - ya) - (yy - ya) * (xb - xa)) / norm
        img += amp * np.exp(-0.5 * (d / (width / 2.355)) ** 2)
        truth = np.abs(d) <= width / 2
    return img.astype(np.float32), truth
