"""TEST INFRASTRUCTURE -- CPU restatement of LA-Cosmic as BlackBOX calls it.

PARITY UNPINNED: the arithmetic lives in the third-party package
``astroscrappy`` (un-pinned in the reference's pyproject.toml:25; the call is
written for "1.0.8 version of astroscrappy", blackbox.py:4319), which is absent
from /root/reference and from this container, and the reference holds no test or
golden vector for it.  This file restates the published algorithm
(van Dokkum 2001, PASP 113, 1420; astroscrappy.detect_cosmics with
sepmed=False, cleantype='medmask', fsmode='median', gain=1, pssl=0,
satlevel=inf) anchored on the reference's call site blackbox.py:4323-4332 and
SURVEY.md Appendix A.1.  Conventions fixed here (and followed bit for bit by
the HIP kernels), all float32:

* median filters of size K leave the outer K//2 rows/columns equal to the input
  (no padding); 3x3 dilation copies the 1-pixel border;
* L+ : 2x2 block replication -> 3x3 Laplacian [[0,-1,0],[-1,4,-1],[0,-1,0]] with
  neighbours outside the (replicated) image dropped -> clip at 0 -> 2x2 mean.
  Evaluation order of one sub-pixel: p = 4*c; p -= right; p -= left; p -= down;
  p -= up (float32 after every step; "down" = next row).  2x2 mean:
  ((tl + tr) + bl) + br, then * 0.25;
* noise = sqrt(max(m5, 1e-5) + float32(rn*rn));  s = L+ / (2*noise);
  sp = s - medfilt5(s);  f = max((m3 - medfilt7(m3)) / noise, 0.01);
* even-count medians (clean_medmask, background level) take the LOWER middle
  element (quick-select index (n-1)//2).

numpy implementation: clear, vectorised, meant for small frames (tests).  The C
twin oracle/lacosmic_c.c is the same algorithm for full-size CPU timing and is
checked against this file in tests/test_lacosmic_oracle.py.
"""
import numpy as np
from scipy import ndimage

F = np.float32


def medfilt(a, k):
    """k x k median, border of k//2 copied from the input (astroscrappy
    PyMedFilt3/5/7)."""
    out = a.copy()
    h = k // 2
    if a.shape[0] > 2 * h and a.shape[1] > 2 * h:
        m = ndimage.median_filter(a, size=k, mode='nearest')
        out[h:-h, h:-h] = m[h:-h, h:-h]
    return out


def dilate3(m):
    """3x3 binary dilation, 1-pixel border copied (astroscrappy PyDilate3)."""
    out = m.copy()
    if m.shape[0] > 2 and m.shape[1] > 2:
        d = ndimage.binary_dilation(m, structure=np.ones((3, 3), bool))
        out[1:-1, 1:-1] = d[1:-1, 1:-1]
    return out


def lplus(clean):
    """L+ of a float32 image (see module docstring for the evaluation order)."""
    ny, nx = clean.shape
    sub = np.repeat(np.repeat(clean, 2, axis=0), 2, axis=1)
    p = F(4) * sub
    # right, left, down, up -- neighbours outside the array are dropped
    p[:, :-1] -= sub[:, 1:]
    p[:, 1:] -= sub[:, :-1]
    p[:-1, :] -= sub[1:, :]
    p[1:, :] -= sub[:-1, :]
    np.maximum(p, F(0), out=p)
    r = ((p[0::2, 0::2] + p[0::2, 1::2]) + p[1::2, 0::2]) + p[1::2, 1::2]
    return (r * F(0.25)).astype(F)


def lower_median(v):
    """quick-select median: element (n-1)//2 of the sorted values"""
    v = np.sort(np.asarray(v, F).ravel())
    return v[(v.size - 1) // 2]


def clean_medmask(clean, crmask, mask, background_level):
    ny, nx = clean.shape
    bad = crmask | mask
    ys, xs = np.nonzero(crmask[2:ny - 2, 2:nx - 2])
    for j, i in zip(ys + 2, xs + 2):
        win = clean[j - 2:j + 3, i - 2:i + 3][~bad[j - 2:j + 3, i - 2:i + 3]]
        clean[j, i] = lower_median(win) if win.size else background_level


def detect_cosmics(indat, inmask, sigclip, sigfrac, objlim, niter, readnoise,
                   return_iters=False):
    """-> (crmask bool, cleanarr float32)   [astroscrappy.detect_cosmics as
    called at blackbox.py:4323-4332]"""
    clean = np.array(indat, dtype=F, order='C', copy=True)
    mask = np.asarray(inmask, bool).copy()
    good = ~mask
    background_level = lower_median(clean[good]) if good.any() else F(0)
    crmask = np.zeros(clean.shape, bool)
    sigclip = F(sigclip)
    sigcliplow = F(F(sigfrac) * sigclip)
    objlim = F(objlim)
    rn = F(readnoise)
    rn2 = F(rn * rn)
    ncr = []
    for _ in range(niter):
        lp = lplus(clean)
        m5 = medfilt(clean, 5)
        m5 = np.maximum(m5, F(0.00001))
        noise = np.sqrt(m5 + rn2).astype(F)
        s = lp / (F(2.0) * noise)
        sp = s - medfilt(s, 5)
        m3 = medfilt(clean, 3)
        f = (m3 - medfilt(m3, 7)) / noise
        f = np.where(f < F(0.01), F(0.01), f)
        with np.errstate(invalid='ignore', divide='ignore'):
            cosmics = (sp > sigclip) & good & ((sp / f) > objlim)
        cosmics = dilate3(cosmics) & good & (sp > sigclip)
        cosmics = dilate3(cosmics) & good & (sp > sigcliplow)
        n = int(cosmics.sum())
        ncr.append(n)
        crmask |= cosmics
        if n == 0:
            break
        clean_medmask(clean, crmask, mask, background_level)
    if return_iters:
        return crmask, clean, ncr
    return crmask, clean


def cosmics_corr(data, data_mask, exptime, sigclip, sigfrac, objlim, niter,
                 readnoise):
    """blackbox.py:4259-4370 -> (data, data_mask, ncosmics_per_sec)"""
    crmask, data = detect_cosmics(data, data_mask != 0, sigclip, sigfrac,
                                  objlim, niter, readnoise)
    data_mask[crmask] |= 2
    ncosmics = ndimage.label(crmask, structure=np.ones((3, 3), bool))[1]
    return data, data_mask, ncosmics / float(exptime)
