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
 SEGMENT: make_mask wants one of the segments detsat returned, [[x0, y0], [x1, y1]] between two edge pixels
   (sat_detect takes the first, blackbox.py:4199).  The probabilistic transform draws them at random; here it
   is the pair of supporting edge pixels of the refined line that lie furthest apart, oriented the way
   skimage's line walk orients its segments (function `trail_segment`); oracle/gen_golden_sat.py runs the real
   probabilistic_hough_line with ten seeds on the golden scenes and tests/test_sat_mask.py holds this
   segment's angle and offset inside the envelope of what it returns.
 MASK (function `make_mask`): acstools.satdet.make_mask(file, ext, segment, sublen=5, pad=0, sigma=5) as
   published, restated from memory of its source [EXT: no copy of acstools exists here] with every
   convention written down below: image / max, negatives to 0; skimage.transform.rotate(image, angle of the
   segment, resize=True, order=3) so that the trail runs along the rows; starting at the segment's first point, windows
   of `subwidth` = 200 columns x 2 `sublen` = 10 rows: median along the trail per row, the rows whose median
   exceeds sigma_clipped_stats(medians)[0] + sigma * biweight_midvariance(medians) are the trail in that
   window; the window centre follows them; first to the right, then to the left of the start, in steps of
   half a window, until the frame ends or a window shows no trail; the strip mask is rotated back
   (order=1, resize=True), cropped about its centre to the frame and made boolean.  The library calls it
   consists of -- skimage.transform.rotate (orders 3 and 1), numpy.median,
   astropy.stats.sigma_clipped_stats, astropy.stats.biweight_midvariance -- are restated on numpy here and
   PINNED against the real ones on the golden scenes (oracle/gen_golden_sat.py -> tests/golden/sat_mask.npz:
   the mask bit for bit, every window's medians to the last bit).  One deliberate difference: the image is
   promoted to float64 before the rotation (scikit-image 0.18 interpolates a float32 image in float32;
   later versions do not) -- the golden masks are the same either way.

What is kept from sat_detect itself: 2x2 SUM binning (4171-4172), one trail at most (the loop is
`range(1)`), `buf=40`, profile threshold `sigma=5` of make_mask, un-binning with np.kron (4224),
`data_mask |= 16`, NSATS = 8-connected label count (4230).

Steps:
 1. binned = 2x2 sums (float32, (a00+a01)+(a10+a11)).
 2. (the largest / smallest binned value: make_mask's normalisation and clip range; reported as info.)
 3. edge map = edges(binned) (front end above).
 4. Hough accumulator: theta_k = radians(2 + k/2), k = 0..351, cell = round_half_away(x cos + y sin)
    + offset, offset = ceil(hypot(ny, nx)) (float64), one vote per edge pixel and theta.
 5. best cell = most votes (ties: smallest (theta, rho) flat index in theta-major order); accepted
    when votes >= 210 and the edge pixels within 1.5 px of that line span the frame: the smallest and
    the largest position along the line both lie within `buf` px of where the line leaves the frame.
 6. segment = trail_segment(supporting edge pixels): the two furthest apart, in skimage's orientation.
 7. mask_binned = make_mask(binned, segment, sublen=5, pad=0, sigma=5); a ValueError of make_mask ("First look
    at finding a profile failed", ...) means no trail is masked (blackbox.py:4204-4210).
 8. np.kron(mask_binned, ones((2, 2))): full-resolution pixels get bit 16; NSATS = 8-connected components.
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


LINE_GAP = 75                                              # acstools: probabilistic_hough_line(..., line_length=200, line_gap=75)


def trail_segment(xs, ys, ct, st, r_index, offset, theta_deg):
    """the segment [[x0, y0], [x1, y1]] handed to make_mask.  skimage's probabilistic Hough transform walks, from a
    random edge pixel, along the raster line of the accumulator's best angle through it, stepping over gaps of up to
    line_gap pixels, and returns the two edge pixels where the walk ended: segments lie along the GRID angle, between
    edge pixels.  The deterministic stand-in: the edge pixels that voted for the best cell (same rounding as the vote),
    ordered along the line (position t = y cos - x sin; ties in raster order); the longest stretch without a gap of
    more than line_gap in t (the first one on ties); its two end pixels -- oriented as skimage orients the ends of its
    walk: for a line that runs more along x (45 < theta < 135 deg) from the smaller x to the larger x; else
    (theta <= 45) from the larger y to the smaller y, (theta >= 135) from the smaller y to the larger y."""
    r = ct * xs + st * ys
    on = (np.where(r > 0, r + 0.5, r - 0.5).astype(np.int64) + offset) == r_index
    px, py = xs[on], ys[on]
    t = py * ct - px * st
    order = np.argsort(t, kind='stable')
    px, py, t = px[order], py[order], t[order]
    cut = np.nonzero(np.diff(t) > LINE_GAP)[0]
    starts = np.concatenate([[0], cut + 1])
    ends = np.concatenate([cut, [t.size - 1]])
    best = int(np.argmax(t[ends] - t[starts]))                      # first maximum
    i0, i1 = int(starts[best]), int(ends[best])
    p, q = (int(px[i0]), int(py[i0])), (int(px[i1]), int(py[i1]))
    if 45.0 < theta_deg < 135.0:
        first_is_p = p[0] < q[0] or (p[0] == q[0] and p[1] <= q[1])
    elif theta_deg <= 45.0:
        first_is_p = p[1] > q[1] or (p[1] == q[1] and p[0] <= q[0])
    else:
        first_is_p = p[1] < q[1] or (p[1] == q[1] and p[0] <= q[0])
    return [list(p), list(q)] if first_is_p else [list(q), list(p)]


# ---- skimage.transform.rotate / warp (orders 3 and 1, mode 'constant', cval 0, clip) on numpy, float64 ----------
def rotate_matrix(shape, dirx, diry):
    """the matrix skimage.transform.rotate(image, angle, resize=True) hands to warp (output (col, row, 1) -> input
    (col, row, 1)) and the output shape, for the angle of the direction (dirx, diry): rotation about
    (cols / 2 - 0.5, rows / 2 - 0.5), the output frame is the bounding box of the rotated corners.  cos and sin are
    dirx / h and diry / h (scikit-image takes them from the angle in degrees: equal to a few 1e-16), every product
    and sum below is one IEEE operation in the written order -- the HIP kernel k_sat_geometry does the very same."""
    rows, cols = shape
    h = float(np.sqrt(np.float64(dirx) * np.float64(dirx) + np.float64(diry) * np.float64(diry)))
    C, S = float(dirx) / h, float(diry) / h
    cx, cy = cols / 2.0 - 0.5, rows / 2.0 - 0.5
    tx = (C * (-cx) + (-S) * (-cy)) + cx                            # T(c) R T(-c)
    ty = (S * (-cx) + C * (-cy)) + cy
    # where the corners of the input go (inverse map: R^T (p - t))
    us, vs = [], []
    for (px, py) in ((0.0, 0.0), (0.0, rows - 1.0), (cols - 1.0, rows - 1.0), (cols - 1.0, 0.0)):
        ax, ay = px - tx, py - ty
        us.append(C * ax + S * ay)
        vs.append((-S) * ax + C * ay)
    minc, maxc, minr, maxr = min(us), max(us), min(vs), max(vs)
    out_rows, out_cols = int(np.around(maxr - minr + 1)), int(np.around(maxc - minc + 1))
    m = np.array([[C, -S, (C * minc + (-S) * minr) + tx], [S, C, (S * minc + C * minr) + ty], [0.0, 0.0, 1.0]])
    return m, (out_rows, out_cols)


def to_output(m, x, y):
    """output position (col, row) that the matrix of rotate_matrix maps onto the input position (x, y)"""
    ax, ay = x - m[0, 2], y - m[1, 2]
    return m[0, 0] * ax + m[1, 0] * ay, m[0, 1] * ax + m[1, 1] * ay


def _cubic(x, f0, f1, f2, f3):
    return f1 + 0.5 * x * (f2 - f0 + x * (2.0 * f0 - 5.0 * f1 + 4.0 * f2 - f3 + x * (3.0 * (f1 - f2) + f3 - f0)))


def _pixel(img, ri, ci):
    rows, cols = img.shape
    ok = (ri >= 0) & (ri < rows) & (ci >= 0) & (ci < cols)
    return np.where(ok, img[np.clip(ri, 0, rows - 1), np.clip(ci, 0, cols - 1)], 0.0)


def warp_rows(img, m, rr, cc, order, lo=None, hi=None):
    """skimage's _warp_fast at the output pixels (rr, cc) (integer arrays): input position by the matrix, bicubic
    (order 3: Catmull-Rom, rows of the 4 x 4 patch first) or bilinear (order 1) interpolation, 0 outside the image;
    then the clipping warp() applies to orders > 0: to [lo, hi] = the input's range, except pixels that are exactly 0"""
    rf, cf = rr.astype(float), cc.astype(float)
    c = m[0, 0] * cf + m[0, 1] * rf + m[0, 2]
    r = m[1, 0] * cf + m[1, 1] * rf + m[1, 2]
    if order == 3:
        r0, c0 = np.floor(r).astype(np.int64), np.floor(c).astype(np.int64)
        xr, xc = r - r0, c - c0
        r0 -= 1; c0 -= 1
        fr = [_cubic(xc, *[_pixel(img, r0 + pr, c0 + pc) for pc in range(4)]) for pr in range(4)]
        out = _cubic(xr, *fr)
    else:
        minr, minc = np.floor(r).astype(np.int64), np.floor(c).astype(np.int64)
        maxr, maxc = np.ceil(r).astype(np.int64), np.ceil(c).astype(np.int64)
        dr, dc = r - minr, c - minc
        top = (1 - dc) * _pixel(img, minr, minc) + dc * _pixel(img, minr, maxc)
        bot = (1 - dc) * _pixel(img, maxr, minc) + dc * _pixel(img, maxr, maxc)
        out = (1 - dr) * top + dr * bot
    if lo is not None:
        keep = out == 0.0 if not (lo <= 0.0 <= hi) else None
        out = np.clip(out, lo, hi)
        if keep is not None:
            out[keep] = 0.0
    return out


def rotate(img, dirx, diry, order):
    """skimage.transform.rotate(img (float64), angle of (dirx, diry), resize=True, order=order)"""
    m, osh = rotate_matrix(img.shape, dirx, diry)
    rr, cc = np.mgrid[0:osh[0], 0:osh[1]]
    return warp_rows(img, m, rr, cc, order, float(img.min()), float(img.max())), m


# ---- astropy.stats.sigma_clipped_stats(x)[0] and biweight_midvariance(x) for short float64 vectors -----------------
def _np_sum(a):
    """numpy's pairwise summation of a contiguous float64 vector of <= 128 elements (np.sum / np.add.reduce)"""
    n = a.size
    if n < 8:
        r = 0.0
        for v in a:
            r = r + v
        return np.float64(r)
    r = [np.float64(a[k]) for k in range(8)]
    i = 8
    while i < n - (n % 8):
        for k in range(8):
            r[k] = r[k] + a[i + k]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        res = res + a[i]
        i += 1
    return np.float64(res)


def sigma_clipped_mean(x, sigma=3.0, maxiters=5):
    """astropy.stats.sigma_clipped_stats(x)[0] (defaults: 3 sigma about the median, std, 5 iterations) in the reference
    environment, where astropy's nan-functions are bottleneck's: mean and std by plain running sums"""
    v = np.asarray(x, np.float64)
    v = v[np.isfinite(v)]

    def seq_mean(a):
        s = 0.0
        for e in a:
            s += e
        return s / a.size

    def seq_std(a):
        mu = seq_mean(a)
        s = 0.0
        for e in a:
            s += (e - mu) * (e - mu)
        return np.sqrt(s / a.size)
    for _ in range(maxiters):
        if v.size == 0:
            break
        cen, sd = np.median(v), seq_std(v)
        keep = (v >= cen - sd * sigma) & (v <= cen + sd * sigma)
        if keep.all():
            break
        v = v[keep]
    return seq_mean(v) if v.size else np.nan


def _pow4(x):
    """x ** 4 by two squarings (numpy calls pow(), whose last bit depends on the libm of the build: the midvariance is
    compared to 1e-14)"""
    x2 = x * x
    return x2 * x2


def biweight_midvariance(x, c=9.0):
    """astropy.stats.biweight_midvariance(x) (c = 9, M = median, sample size = all points)"""
    d0 = np.asarray(x, np.float64)
    med = np.median(d0)
    d = d0 - med
    mad = np.median(np.abs(d0 - med))
    if mad == 0.0:
        return 0.0
    u = d / (c * mad)
    inside = np.abs(u) < 1
    u = u * u
    f1 = d * d * _pow4(1.0 - u)
    f1[~inside] = 0.0
    f2 = (1.0 - u) * (1.0 - 5.0 * u)
    f2[~inside] = 0.0
    return float(d0.size * _np_sum(f1) / (np.abs(_np_sum(f2)) ** 2))


def valid_indices(shape, ix0, ix1, iy0, iy1):
    """acstools.satdet._get_valid_indices: clip to the array, integer part; IndexError when nothing is left"""
    ymax, xmax = shape
    ix0, iy0 = max(ix0, 0), max(iy0, 0)
    ix1, iy1 = min(ix1, xmax), min(iy1, ymax)
    if iy1 <= iy0 or ix1 <= ix0:
        raise IndexError('array[{}:{},{}:{}] is invalid'.format(iy0, iy1, ix0, ix1))
    return int(ix0), int(ix1), int(iy0), int(iy1)


def profile_rows(medarr, sigma):
    """rows of a window that belong to the trail: median above clipped mean + sigma * biweight midvariance -> indices"""
    mean = sigma_clipped_mean(medarr)
    var = biweight_midvariance(medarr)
    return np.nonzero(medarr > mean + sigma * var)[0], mean, var


def make_mask(image, segment, sublen=5, subwidth=200, sigma=5.0, pad=0, return_debug=False, prims=None):
    """acstools.satdet.make_mask as called by sat_detect (see the module docstring) -> bool mask of the image's shape.
    prims: the library functions instead of their restatements here (oracle/gen_golden_sat.py passes scikit-image's
    rotate and astropy's statistics): dict(rotate3=f(img64, deg) -> rotated, rows=f(medarr, sigma) -> (z, mean, var),
    rotate1=f(mask, deg) -> rotated back)"""
    image = np.asarray(image, F)
    top = image.max()
    if not top > 0:
        raise ValueError('Image has no positive values')
    img = (image / top).astype(F)
    img[img < 0] = 0
    (x0, y0), (x1, y1) = segment
    deg = np.degrees(np.arctan2(float(y1 - y0), float(x1 - x0)))
    ddx, ddy = float(x1 - x0), float(y1 - y0)
    m, _ = rotate_matrix(img.shape, ddx, ddy)
    rot = prims['rotate3'](img.astype(np.float64), deg) if prims else rotate(img.astype(np.float64), ddx, ddy, 3)[0]
    # the start point in the rotated frame: the output position the matrix maps onto (x0, y0)
    sx, sy = to_output(m, float(x0), float(y0))
    dx = int(subwidth / 2)
    mask = np.zeros(rot.shape)
    windows = []

    def look(cx, cy):
        ix0, ix1, iy0, iy1 = valid_indices(rot.shape, cx - dx, cx + dx, cy - sublen, cy + sublen)
        sub = rot[iy0:iy1, ix0:ix1]
        medarr = np.median(sub, axis=1)
        z, mean, var = prims['rows'](medarr, sigma) if prims else profile_rows(medarr, sigma)
        windows.append(dict(box=(ix0, ix1, iy0, iy1), medarr=medarr, mean=mean, var=var, z=z.tolist()))
        return (ix0, ix1, iy0, iy1), z, sub
    (ix0, ix1, iy0, iy1), z, sub = look(sx, sy)
    if len(sub) <= sublen:
        raise ValueError('Trail subarray size is {} but expected {} or larger'.format(len(sub), sublen))
    if len(z) < 1:
        raise ValueError('First look at finding a profile failed. Nothing found at {} from background!'.format(sigma))

    def paint(box, z):
        lower, upper = int(z.min()) - pad, int(z.max()) + pad
        r0, r1 = max(box[2] + lower, 0), min(box[2] + upper + 1, rot.shape[0])
        mask[r0:r1, box[0]:box[1]] = 1
        return box[2] + int(np.ceil(z.min() + (z.max() - z.min()) / 2.0))
    centre0 = paint((ix0, ix1, iy0, iy1), z)
    for direction in (1, -1):
        nextx, centre = sx + direction * dx, centre0
        for _ in range(500):
            try:
                box, z, sub = look(nextx, centre)
            except IndexError:
                break                                              # the window left the frame
            if len(z) < 1:
                break                                              # no trail in this window
            centre = paint(box, z)
            nextx += direction * dx
    m2, osh2 = rotate_matrix(mask.shape, ddx, -ddy)
    ny, nx = image.shape
    iy0, ix0 = int((osh2[0] - ny) / 2), int((osh2[1] - nx) / 2)
    if prims:
        back = prims['rotate1'](mask, -deg)
        assert back.shape == osh2
        back = back[iy0:iy0 + ny, ix0:ix0 + nx]
    else:
        rr, cc = np.mgrid[iy0:iy0 + ny, ix0:ix0 + nx]
        back = warp_rows(mask, m2, rr, cc, 1)
    out = back.astype(bool)
    return (out, dict(windows=windows, deg=deg, start=(sx, sy), rot_shape=rot.shape)) if return_debug else out


def detect(data, return_debug=False):
    """-> (mask_sat uint8 full resolution, nsats, info dict)"""
    b = bin2(data)
    ny, nx = b.shape
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
    info = dict(bmax=float(b[np.isfinite(b)].max()), bmin=float(b[np.isfinite(b)].min()), nedge=int(edge.sum()), votes=votes, theta=theta,
                rho=rho, k=k)
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
    seg = trail_segment(xs, ys, ct[k], st[k], r, off, THETA_DEG[k])
    if seg[0] == seg[1]:
        return (mask_full, 0, info)
    info['segment'] = seg
    try:
        mask_binned, dbg = make_mask(b, seg, return_debug=True)
    except ValueError as e:                                        # "satellite trail found but could not be fitted"
        info['make_mask_error'] = str(e)
        return (mask_full, 0, info)
    info['windows'] = dbg['windows']
    mask_binned = mask_binned.astype(np.uint8)
    mask_full = np.kron(mask_binned, np.ones((2, 2), np.uint8)).astype(np.uint8)
    nsats = ndimage.label(mask_full, structure=np.ones((3, 3), bool))[1]
    return (mask_full, int(nsats), info)


def sat_detect(data, data_mask):
    """blackbox.py:4163-4254 -> (data_mask with bit 16 added, NSATS)"""
    m, nsats, info = detect(data)
    data_mask[m == 1] |= 16
    return data_mask, nsats, info
