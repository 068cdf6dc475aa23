"""a7 nonlin_corr (blackbox.py:7394-7437).  CPU: the de Boor evaluation the HIP kernel uses
(FITPACK splev/fpbspl, restated here in numpy with the same operation order) equals scipy's
own spline call bit for bit; the oracle keeps the reference's quirks.  GPU: the HIP kernel
equals the oracle bit for bit, standalone and inside the fused calibration pass."""
import numpy as np
import pytest
from scipy import interpolate

import bbx_oracle as O
from blackbox_amd import settings, synth


def make_splines(seed=3):
    rs = np.random.RandomState(seed)
    out = []
    x = np.linspace(0, 62000, 400)
    for c in range(16):
        y = (0.004 + 0.002 * rs.rand()) * (x / 6e4) ** 2 - 0.003 * (x / 6e4) + 2e-4 * rs.normal(size=x.size)
        out.append(interpolate.UnivariateSpline(x, y, k=3, s=400 * (2e-4) ** 2 * (0.8 + 0.4 * rs.rand())))
    return out


def splev_deboor(t, c, k, x):
    """splev.f / fpbspl.f (scipy 1.7), float64, one x at a time"""
    t = np.asarray(t, np.float64); c = np.asarray(c, np.float64)
    n, k1 = t.size, k + 1
    nk1 = n - k1
    out = np.empty(len(x))
    for idx, arg in enumerate(np.asarray(x, np.float64)):
        l = k1
        while not (arg < t[l]) and l != nk1:
            l += 1
        h = np.zeros(k1 + 1); h[0] = 1.0
        for j in range(1, k + 1):
            hh = h[:j].copy()
            h[0] = 0.0
            for i in range(1, j + 1):
                li = l + i; lj = li - j
                if t[li - 1] == t[lj - 1]:
                    h[i] = 0.0
                    continue
                f = hh[i - 1] / (t[li - 1] - t[lj - 1])
                h[i - 1] = h[i - 1] + f * (t[li - 1] - arg)
                h[i] = f * (arg - t[lj - 1])
        sp = 0.0
        for j in range(k1):
            sp = sp + c[l - k1 + j] * h[j]
        out[idx] = sp
    return out


def test_deboor_equals_scipy_bitwise():
    rs = np.random.RandomState(0)
    for spl in make_splines()[:4]:
        t, c, k = spl._eval_args
        x = np.concatenate([rs.uniform(-500, 50000, 3000), t, [0.0, 50000.0, -1e4, 7e4]]).astype(np.float32)
        assert np.array_equal(splev_deboor(t, c, k, x), spl(x))


def test_oracle_quirks():
    spl = make_splines()
    ys, xs = 8, 12
    data = np.full((2 * ys, 8 * xs), 1000.0, np.float32)
    data[0, 0] = 3e5                                  # counts > 50000: "uncorrected" -> halved (sic)
    gain = settings.gain['ML1']
    out = O.nonlin_corr(data.copy(), spl, gain, ys, xs)
    assert out[0, 0] == np.float32(3e5 / 2)
    c0 = np.float32(1000.0) / np.float32(gain[0])
    assert out[1, 1] == np.float32(np.float64(np.float32(1000.0)) / (spl[0](np.array([c0], np.float32))[0] + 1))


@pytest.mark.gpu
def test_gpu_nonlin_bitexact():
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from blackbox_amd import reduce as R
    tel, ys, xs = 'ML1', 96, 132
    spl = make_splines()
    ctx = R.Context(0)
    rs = np.random.RandomState(1)
    data = rs.uniform(-200, 160000, (2 * ys, 8 * xs)).astype(np.float32)
    data[3, 5] = np.nan
    geom = R.geometry((2 * (ys + 20), 8 * (xs + 45)), ys, xs)
    d = torch.from_numpy(data.copy()).to(ctx.device)
    R.nonlin_corr(ctx, d, geom, tel, splines=spl)
    ctx.sync()
    want = O.nonlin_corr(data.copy(), spl, settings.gain[tel], ys, xs)
    assert np.array_equal(d.cpu().numpy(), want, equal_nan=True)
    R.set_nonlin(ctx, None)

    # inside the fused calibration: reduce_object(nonlin_splines=...) == oracle chain
    ys, xs = 64, 330
    case = synth.make_case(ys, xs, 21, tel=tel, os_y=20, os_x=45, n_stars=30, n_sat=2, n_cr=0)
    raw = torch.from_numpy(case['raw']).to(ctx.device)
    flat = torch.from_numpy(case['flat']).to(ctx.device)
    bpm = torch.from_numpy(case['bpm']).to(ctx.device)
    got, gmask, _, _ = R.reduce_object(ctx, raw, {}, tel, mflat=flat, bpm=bpm, ysize_chan=ys, xsize_chan=xs,
                                       do_cosmics=False, detect_sats=False, nonlin_splines=spl)
    gain, sat = settings.gain[tel], settings.satlevel[tel]
    o = case['raw'].astype(np.float32)
    O.gain_corr(o, gain, ys, xs)
    o, oh, _ = O.os_corr(o, ys, xs, tel=tel, accum='bn32')
    O.nonlin_corr(o, spl, gain, ys, xs)
    omask, _ = O.mask_init(o, oh, case['bpm'], gain, sat, ys, xs)
    o /= case['flat']
    O.edge_fill(o, omask, ys, xs)
    assert np.array_equal(gmask.cpu().numpy(), omask)
    assert np.array_equal(got.cpu().numpy(), o)
    ctx.close()
