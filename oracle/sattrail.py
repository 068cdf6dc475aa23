"""TEST INFRASTRUCTURE -- CPU restatement of the satellite-trail masking step.

PARITY UNPINNED, and not pinnable: the reference runs either ASTA (a Keras U-Net whose
weights `model-best.keras` are not available, blackbox.py:4090-4158) or
acstools.satdet.detsat/make_mask (blackbox.py:4163-4254), whose probabilistic Hough
transform draws random pixel subsets -- the reference's own output is not deterministic.
What is kept from sat_detect: 2x2 SUM binning (4171-4172), one trail at most (the loop is
`range(1)`), detection significance `sigma=3`, line support threshold `h_thresh=0.2` of the
chord, edge buffer `buf=40`, profile threshold `sigma=5` of make_mask, un-binning with
np.kron (4224), `data_mask |= 16`, NSATS = 8-connected label count (4230).

Deterministic classical detector fixed here (and followed by the HIP kernels):
 1. binned = 2x2 sums (float32, (a00+a01)+(a10+a11)).
 2. level, sigma = 3-sigma clipped mean / std of the binned frame (4 passes: all finite
    pixels, then three clips), float64.
 3. edge pixels: level + 3 sigma < value < level + 50 sigma (star cores are left out).
 4. Hough accumulator: theta_k = k*pi/NTHETA (NTHETA = 720), rho = x cos + y sin rounded to
    the nearest integer (float64 math), one vote per edge pixel and theta.
 5. best cell = most votes (ties: smallest flat index); accepted when votes >= 200 and
    votes >= 0.2 * chord length of that line inside the binned frame, and the chord reaches
    within `buf` = 40 px of the frame border at both ends (always true for a chord).
 6. perpendicular profile P(d), d = -40..40 (binned px): mean of the non-star pixels
    (value < level + 50 sigma) whose rounded signed distance to the line is d.
 7. strip = the contiguous run of offsets around the profile maximum with
    P(d) - level > max(5 sigma / sqrt(n_d), 0.1 (P_max - level)); a trail needs >= 1 offset.
 8. full-resolution pixels whose binned cell centre lies in the strip get bit 16.
"""
import numpy as np
from scipy import ndimage

NTHETA = 720
PROF_HALF = 40
F = np.float32


def bin2(data):
    d = data.astype(F)
    return ((d[0::2, 0::2] + d[0::2, 1::2]) + (d[1::2, 0::2] + d[1::2, 1::2])).astype(F)


def clipped_level(b):
    v = b[np.isfinite(b)].astype(np.float64)
    lo, hi = -np.inf, np.inf
    mean = std = 0.0
    for _ in range(4):
        w = v[(v >= lo) & (v <= hi)]
        mean = w.sum() / w.size
        std = np.sqrt(max((w * w).sum() / w.size - mean * mean, 0.0))
        lo, hi = max(lo, mean - 3 * std), min(hi, mean + 3 * std)
    return mean, std


def chord_length(theta, rho, ny, nx):
    """length of the line x cos + y sin = rho inside [0, nx-1] x [0, ny-1]"""
    c, s = np.cos(theta), np.sin(theta)
    pts = []
    for x in (0.0, nx - 1.0):
        if abs(s) > 1e-12:
            y = (rho - x * c) / s
            if -1e-9 <= y <= ny - 1 + 1e-9:
                pts.append((x, y))
    for y in (0.0, ny - 1.0):
        if abs(c) > 1e-12:
            x = (rho - y * s) / c
            if -1e-9 <= x <= nx - 1 + 1e-9:
                pts.append((x, y))
    if len(pts) < 2:
        return 0.0
    best = 0.0
    for i in range(len(pts)):
        for j in range(i + 1, len(pts)):
            best = max(best, np.hypot(pts[i][0] - pts[j][0], pts[i][1] - pts[j][1]))
    return best


def detect(data, return_debug=False):
    """-> (mask_sat uint8 full resolution, nsats, info dict)"""
    b = bin2(data)
    ny, nx = b.shape
    level, sigma = clipped_level(b)
    bd = b.astype(np.float64)
    edge = (bd > level + 3 * sigma) & (bd < level + 50 * sigma)
    ys, xs = np.nonzero(edge)
    nrho = 2 * int(np.ceil(np.hypot(ny, nx))) + 1
    off = nrho // 2
    acc = np.zeros((NTHETA, nrho), np.int64)
    th = np.arange(NTHETA) * (np.pi / NTHETA)
    ct, st = np.cos(th), np.sin(th)
    for k in range(NTHETA):
        r = np.floor(xs * ct[k] + ys * st[k] + 0.5).astype(np.int64) + off
        np.add.at(acc[k], r, 1)
    flat = int(np.argmax(acc))
    votes = int(acc.reshape(-1)[flat])
    k, r = divmod(flat, nrho)
    theta, rho = th[k], float(r - off)
    info = dict(level=level, sigma=sigma, nedge=int(edge.sum()), votes=votes, theta=theta, rho=rho)
    mask_full = np.zeros(data.shape, np.uint8)
    chord = chord_length(theta, rho, ny, nx)
    info['chord'] = chord
    if votes < 200 or votes < 0.2 * chord:
        return (mask_full, 0, info)
    yy, xx = np.mgrid[0:ny, 0:nx]
    dist = np.floor(xx * ct[k] + yy * st[k] - rho + 0.5).astype(np.int64)
    use = (np.abs(dist) <= PROF_HALF) & (bd < level + 50 * sigma) & np.isfinite(bd)
    prof_sum = np.zeros(2 * PROF_HALF + 1)
    prof_n = np.zeros(2 * PROF_HALF + 1)
    np.add.at(prof_sum, dist[use] + PROF_HALF, bd[use])
    np.add.at(prof_n, dist[use] + PROF_HALF, 1)
    with np.errstate(invalid='ignore', divide='ignore'):
        prof = np.where(prof_n > 0, prof_sum / prof_n, level) - level
        thr = np.maximum(5 * sigma / np.sqrt(np.maximum(prof_n, 1)), 0.1 * prof.max())
    ipk = int(np.argmax(prof))
    above = (prof > thr) & (prof_n > 0)
    if not above[ipk]:
        return (mask_full, 0, info)
    lo = ipk
    while lo - 1 >= 0 and above[lo - 1]:
        lo -= 1
    hi = ipk
    while hi + 1 < above.size and above[hi + 1]:
        hi += 1
    info['strip'] = (lo - PROF_HALF, hi - PROF_HALF)
    mask_binned = ((dist >= lo - PROF_HALF) & (dist <= hi - PROF_HALF)).astype(np.uint8)
    mask_full = np.kron(mask_binned, np.ones((2, 2), np.uint8)).astype(np.uint8)
    nsats = ndimage.label(mask_full, structure=np.ones((3, 3), bool))[1]
    return (mask_full, int(nsats), info)


def sat_detect(data, data_mask):
    """blackbox.py:4163-4254 -> (data_mask with bit 16 added, NSATS)"""
    m, nsats, info = detect(data)
    data_mask[m == 1] |= 16
    return data_mask, nsats, info
