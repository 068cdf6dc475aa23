"""TEST INFRASTRUCTURE -- CPU restatement of the satellite-trail masking step.

PARITY: the reference runs either ASTA (a Keras U-Net whose weights `model-best.keras` are not
available, blackbox.py:4090-4158) or acstools.satdet.detsat / make_mask (blackbox.py:4163-4254;
acstools is not in this image and the reference holds no fixture for it).  acstools' detector is, as
published: percentile (4.5, 93) intensity rescale -> skimage Canny (sigma, thresholds low_thresh /
h_thresh of the image maximum) -> remove_small_objects(60) -> *probabilistic* Hough transform over
theta = 2 .. 177.5 deg in half-degree steps (threshold 210 votes, random pixel order: its output is
not deterministic) -> lines that reach the image borders within `buf` -> make_mask (profile across the
trail, strip above `sigma`).  What is restated here, and how it is pinned:

 FRONT END (function `edges`): exactly those library calls with the arguments sat_detect passes
   (sigma=3, h_thresh=0.2, low_thresh default 0.1), restated on numpy / scipy.ndimage and PINNED
   bit for bit against scikit-image 0.18.3 run in the build container (oracle/gen_golden_sat.py ->
   tests/golden/sat_front.npz: percentiles, rescaled image, Canny edge map, map after
   remove_small_objects).
 HOUGH: the full accumulator over acstools' theta grid with skimage.transform.hough_line's cell
   definition (PINNED against it the same way) stands in for the probabilistic transform; the
   best cell needs acstools' 210 votes and its supporting edge pixels must reach within
   `buf` = 40 px of the frame border at both ends (acstools keeps lines that end near the borders).
 MASK: this repository's own profile strip (steps 6-8 below), not acstools' rotate-and-profile
   make_mask -- PARITY UNPINNED for that part.

What is kept from sat_detect itself: 2x2 SUM binning (4171-4172), one trail at most (the loop is
`range(1)`), `buf=40`, profile threshold `sigma=5` of make_mask, un-binning with np.kron (4224),
`data_mask |= 16`, NSATS = 8-connected label count (4230).

Steps:
 1. binned = 2x2 sums (float32, (a00+a01)+(a10+a11)).
 2. level, sigma = 3-sigma clipped mean / std of the binned frame (4 passes), float64 -- used by the
    profile only.
 3. edge map = edges(binned) (front end above).
 4. Hough accumulator: theta_k = radians(2 + k/2), k = 0..351, cell = round_half_away(x cos + y sin)
    + offset, offset = ceil(hypot(ny, nx)) (float64), one vote per edge pixel and theta.
 5. best cell = most votes (ties: smallest (theta, rho) flat index in theta-major order); accepted
    when votes >= 210 and the edge pixels within 1.5 px of that line span the frame: the smallest and
    the largest position along the line both lie within `buf` px of where the line leaves the frame.
 6. perpendicular profile P(d), d = -40..40 (binned px): mean of the non-star pixels
    (value < level + 50 sigma) whose rounded signed distance to the line is d.
 7. strip = the contiguous run of offsets around the profile maximum with
    P(d) - level > max(5 sigma / sqrt(n_d), 0.1 (P_max - level)); a trail needs >= 1 offset.
 8. full-resolution pixels whose binned cell centre lies in the strip get bit 16.
"""
import numpy as np
from scipy import ndimage

THETA_DEG = np.arange(2, 178, 0.5, dtype=float)          # acstools.satdet: np.radians(np.arange(2, 178, 0.5))
NTHETA = THETA_DEG.size
PROF_HALF = 40
BUF = 40
MIN_VOTES = 210
CANNY_SIGMA, LOW_THRESH, HIGH_THRESH, SMALL_EDGE = 3.0, 0.1, 0.2, 60
PERCENTILES = (4.5, 93.0)
REFINE_TOL = (None, 3.0, 2.0, 1.5)                        # rounds of the line refinement: cone, then residual limits [px]
TAN_HALF_STEP = 0.004363350820701567                      # tan(0.25 deg): half a step of the theta grid
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


def order_percentile(sorted_vals, q):
    """numpy.percentile's default (linear) rule on sorted float32 values -> float64:
    virtual index (n-1) q / 100, the float32 difference of the two neighbours, then
    a + diff * g (g < 0.5) or b - diff * (1 - g)"""
    n = sorted_vals.size
    v = (n - 1) * (q / 100.0)
    lo = int(np.floor(v))
    hi = min(lo + 1, n - 1)
    g = v - lo
    a, b = sorted_vals[lo], sorted_vals[hi]
    diff = np.float64(np.float32(b - a))
    return np.float64(a) + diff * g if g < 0.5 else np.float64(b) - diff * (1.0 - g)


def rescale(b):
    """np.percentile(image, (4.5, 93)); p1 < 0 -> 0; skimage.exposure.rescale_intensity(image,
    in_range=(p1, p2)) on a float32 image: clip, subtract, divide -- in float32 with the limits
    rounded to float32 and the float64 difference p2 - p1 rounded to float32"""
    sv = np.sort(b.reshape(-1))
    p1, p2 = order_percentile(sv, PERCENTILES[0]), order_percentile(sv, PERCENTILES[1])
    if p1 < 0:
        p1 = 0.0
    p1f, p2f, den = F(p1), F(p2), F(p2 - p1)
    img = ((np.clip(b, p1f, p2f) - p1f) / den).astype(F) if p1 != p2 else np.clip(b, p1f, p2f)
    return img, p1, p2


def gauss_weights(sigma=CANNY_SIGMA, truncate=4.0):
    """scipy.ndimage's Gaussian kernel: radius int(truncate sigma + 0.5), exp(-0.5 / sigma^2 x^2) normalised"""
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1)
    w = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return w / w.sum()


def canny_fields(img, sigma=CANNY_SIGMA):
    """the smoothed image and gradient fields of skimage.feature.canny(image, sigma) without a mask:
    Gaussian of the float32 image (zero outside, float32 result) divided by the Gaussian of ones
    (float64) + eps; Sobel derivatives (reflecting borders) and their magnitude, float64"""
    sm32 = ndimage.gaussian_filter(img, sigma, mode='constant', cval=0, output=F)
    bleed = ndimage.gaussian_filter(np.ones(img.shape, np.float64), sigma, mode='constant', cval=0)
    sm = sm32 / (bleed + np.finfo(float).eps)
    js = ndimage.sobel(sm, axis=1)
    is_ = ndimage.sobel(sm, axis=0)
    mag = np.sqrt(is_ * is_ + js * js)                          # np.hypot in skimage: equal up to the last bit
    return sm, is_, js, mag


def canny(img, low, high, sigma=CANNY_SIGMA):
    """skimage.feature.canny(img, sigma, low_threshold=low, high_threshold=high): non-maximum
    suppression with linear interpolation between the two neighbours nearest to the gradient
    direction (four direction sectors; where sectors overlap -- equal absolute derivatives -- the later
    sector's verdict stands), hysteresis: 8-connected components of the low mask that hold a
    pixel of the high mask"""
    sm, is_, js, mag = canny_fields(img, sigma)
    ny, nx = img.shape
    ai, aj = np.abs(is_), np.abs(js)
    inner = np.zeros((ny, nx), bool)
    inner[1:-1, 1:-1] = True                                       # binary_erosion of the all-ones mask, border 0
    inner &= mag > 0

    def sh(dy, dx):
        """mag at (y + dy, x + dx), 0 outside (never used there: inner pixels only)"""
        out = np.zeros_like(mag)
        ys = slice(max(dy, 0), ny + min(dy, 0)); yd = slice(max(-dy, 0), ny + min(-dy, 0))
        xs = slice(max(dx, 0), nx + min(dx, 0)); xd = slice(max(-dx, 0), nx + min(-dx, 0))
        out[yd, xd] = mag[ys, xs]
        return out
    local = np.zeros((ny, nx), bool)
    with np.errstate(divide='ignore', invalid='ignore'):
        w_ji, w_ij = aj / ai, ai / aj
    same = ((is_ >= 0) & (js >= 0)) | ((is_ <= 0) & (js <= 0))
    opp = ((is_ <= 0) & (js >= 0)) | ((is_ >= 0) & (js <= 0))
    # (sector mask, weight, (c1, c2) on the plus side, (c1, c2) on the minus side): c2 w + c1 (1 - w) <= m
    sectors = [(same & (ai >= aj), w_ji, ((1, 0), (1, 1)), ((-1, 0), (-1, -1))),
               (same & (ai <= aj), w_ij, ((0, 1), (1, 1)), ((0, -1), (-1, -1))),
               (opp & (ai <= aj), w_ij, ((0, 1), (-1, 1)), ((0, -1), (1, -1))),
               (opp & (ai >= aj), w_ji, ((-1, 0), (-1, 1)), ((1, 0), (1, -1)))]
    for sel, w, plus, minus in sectors:
        sel = sel & inner
        with np.errstate(invalid='ignore'):
            cp = sh(*plus[1]) * w + sh(*plus[0]) * (1 - w) <= mag
            cm = sh(*minus[1]) * w + sh(*minus[0]) * (1 - w) <= mag
        local[sel] = (cp & cm)[sel]
    low_mask = local & (mag >= low)
    high_mask = local & (mag >= high)
    lab, n = ndimage.label(low_mask, structure=np.ones((3, 3), bool))
    if n == 0:
        return low_mask
    has = np.zeros(n + 1, bool)
    has[np.unique(lab[high_mask])] = True
    has[0] = False
    return has[lab]


def remove_small(edge, min_size=SMALL_EDGE):
    """skimage.morphology.remove_small_objects(edge, min_size, connectivity=8)"""
    lab, n = ndimage.label(edge, structure=np.ones((3, 3), bool))
    size = np.bincount(lab.reshape(-1))
    keep = size >= min_size
    keep[0] = False
    return keep[lab]


def edges(b, return_all=False):
    """the acstools front end on a binned frame -> boolean edge map"""
    img, p1, p2 = rescale(b)
    immax = np.float64(img.max())
    e = canny(img, immax * LOW_THRESH, immax * HIGH_THRESH)
    k = remove_small(e)
    return (k, e, img, p1, p2) if return_all else k


def hough(edge):
    """skimage.transform.hough_line(edge, theta=radians(THETA_DEG)) -> (accumulator [nrho][ntheta], offset)"""
    ny, nx = edge.shape
    offset = int(np.ceil(np.sqrt(ny * ny + nx * nx)))
    nrho = 2 * offset
    th = np.radians(THETA_DEG)
    ct, st = np.cos(th), np.sin(th)
    ys, xs = np.nonzero(edge)
    acc = np.zeros((nrho, NTHETA), np.int64)
    for k in range(NTHETA):
        r = ct[k] * xs + st[k] * ys
        idx = np.where(r > 0, r + 0.5, r - 0.5).astype(np.int64) + offset     # round half away from zero, truncating cast
        np.add.at(acc[:, k], idx, 1)
    return acc, offset


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


def chord_span(c, s, rho, ny, nx):
    """positions t = y c - x s along the line x c + y s = rho where it crosses the frame rectangle
    [0, nx-1] x [0, ny-1] -> (t0, t1) or None"""
    ts = []
    for x in (0.0, nx - 1.0):
        if abs(s) > 1e-12:
            y = (rho - x * c) / s
            if -1e-9 <= y <= ny - 1 + 1e-9:
                ts.append(y * c - x * s)
    for y in (0.0, ny - 1.0):
        if abs(c) > 1e-12:
            x = (rho - y * s) / c
            if -1e-9 <= x <= nx - 1 + 1e-9:
                ts.append(y * c - x * s)
    return (min(ts), max(ts)) if len(ts) >= 2 else None


def detect(data, return_debug=False):
    """-> (mask_sat uint8 full resolution, nsats, info dict)"""
    b = bin2(data)
    ny, nx = b.shape
    level, sigma = clipped_level(b)
    bd = b.astype(np.float64)
    edge = edges(b)
    ys, xs = np.nonzero(edge)
    acc, off = hough(edge)
    nrho = acc.shape[0]
    th = np.radians(THETA_DEG)
    ct, st = np.cos(th), np.sin(th)
    flat = int(np.argmax(acc.T))                                  # theta-major: first maximum
    k, r = divmod(flat, nrho)
    votes = int(acc[r, k])
    theta, rho = th[k], float(r - off)
    info = dict(level=level, sigma=sigma, nedge=int(edge.sum()), votes=votes, theta=theta, rho=rho, k=k)
    mask_full = np.zeros(data.shape, np.uint8)
    info['chord'] = chord_length(theta, rho, ny, nx)
    span = chord_span(ct[k], st[k], rho, ny, nx)
    if votes < MIN_VOTES or span is None:
        return (mask_full, 0, info)
    # the line of the best cell is known to half a grid step (0.25 deg) only: over a 5000-px frame that
    # is several pixels at the ends.  Refine it on the edge pixels inside a cone of that opening around
    # it, then three more rounds on the pixels within 3, 2, 1.5 px of the line fitted so far (a trail has
    # two edges: the rounds settle on one) -- least squares d = a + b t on fixed-point (1/16 px)
    # coordinates, integer sums, so that the result does not depend on the order of the pixels -- like
    # acstools takes the trail's position from the end points of the segments it found, not from the
    # accumulator cell
    d = ct[k] * xs + st[k] * ys - rho
    t = ys * ct[k] - xs * st[k]
    tm = 0.5 * (span[0] + span[1])
    icpt, slope = 0.0, 0.0
    for it, tol in enumerate(REFINE_TOL):
        # first round: the cone around the cell's line; then the pixels near the line fitted so far
        sel = (np.abs(d) <= 1.5 + TAN_HALF_STEP * np.abs(t - tm)) if it == 0 else (np.abs(d - (icpt + slope * t)) <= tol)
        n = int(sel.sum())
        if n < 2:
            return (mask_full, 0, info)
        tq = np.floor(t[sel] * 16.0 + 0.5).astype(np.int64)
        dq = np.floor(d[sel] * 16.0 + 0.5).astype(np.int64)
        St, Sd, Stt, Std = float(tq.sum()), float(dq.sum()), float((tq * tq).sum()), float((tq * dq).sum())
        mt, md = St / n, Sd / n
        var, cov = Stt / n - mt * mt, Std / n - mt * md
        slope = cov / var if var > 0 else 0.0
        icpt = (md - slope * mt) / 16.0
    info['refine'] = (icpt, slope, n)
    # the supporting edge pixels (within 1.5 px of the refined line) must span the frame: acstools keeps
    # lines that end near the borders
    sup = np.abs(d - (icpt + slope * t)) <= 1.5
    if not sup.any():
        return (mask_full, 0, info)
    info['support'] = (float(t[sup].min()), float(t[sup].max()), span)
    if not (t[sup].min() - span[0] <= BUF and span[1] - t[sup].max() <= BUF):
        return (mask_full, 0, info)
    yy, xx = np.mgrid[0:ny, 0:nx]
    dfull = xx * ct[k] + yy * st[k] - rho
    tfull = yy * ct[k] - xx * st[k]
    dist = np.floor(dfull - (icpt + slope * tfull) + 0.5).astype(np.int64)
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
