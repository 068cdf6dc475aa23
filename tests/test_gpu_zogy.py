"""GPU: background mesh, ZOGY sub-image subtraction (rocFFT) and PSF photometry against the
CPU restatement oracle/zogy_core.py (parity unpinned: zogy is absent)
plus the property pins of that section: identical new/ref -> D == 0, pure noise ->
Scorr ~ N(0,1) within the QC ranges (set_qc.py:382-383), injected source -> Fpsf = flux."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
if not torch.cuda.is_available():
    pytest.skip('no GPU', allow_module_level=True)

import zogy_core as Z                       # noqa: E402
from blackbox_amd import reduce as R       # noqa: E402
from blackbox_amd import zogy as G          # noqa: E402

F = np.float32


@pytest.fixture(scope='module')
def ctx():
    c = R.Context(0)
    yield c
    c.close()


def dev(ctx, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)


@pytest.mark.parametrize('box', [2, 5, 33, 60, 64])
def test_box_statistics_box_sizes(ctx, box):
    """k_bkg_boxstats sorts a box in the registers of one wave (64 x 64 slots): box sizes that
    fill them partly, exactly and barely; negative and tied values; constant boxes (std 0);
    boxes below limfrac; outliers on both sides."""
    from blackbox_amd._lib import lib, check
    import ctypes as C
    rs = np.random.RandomState(box)
    nby, nbx = (9, 11) if box > 8 else (40, 37)
    ny, nx = nby * box, nbx * box
    data = np.round(rs.normal(-3, 20, (ny, nx)) * 4) / 4          # quarter-ADU grid: many ties, both signs
    data[rs.random_sample((ny, nx)) < 0.02] += 5000
    data[rs.random_sample((ny, nx)) < 0.02] -= 7000
    data = data.astype(F)
    data[0:box, 0:box] = 17.25                                     # constant box
    data[box:2 * box, 0:box] = np.where(rs.random_sample((box, box)) < 0.5, F(1), F(2))   # two values
    mask = np.zeros((ny, nx), np.uint8)
    mask[rs.random_sample((ny, nx)) < 0.1] = 2
    mask[2 * box:3 * box, box:2 * box][rs.random_sample((box, box)) < 0.6] = 1            # below limfrac
    mask[3 * box:4 * box, 0:box] = 8                               # nothing usable
    objmask = (rs.random_sample((ny, nx)) < 0.05).astype(np.uint8)
    for om, od in ((objmask, dev(ctx, objmask)), (None, None)):
        med_o, std_o = Z.get_back_mini(data, mask, om, box=box)
        m = torch.full((nby, nbx), -1.0, dtype=torch.float32, device=ctx.device)
        s = torch.full((nby, nbx), -1.0, dtype=torch.float32, device=ctx.device)
        t_data, t_mask = dev(ctx, data), dev(ctx, mask)
        check(lib.bbx_bkg_boxstats(ctx.h, ny, nx, box, C.c_void_p(t_data.data_ptr()), C.c_void_p(t_mask.data_ptr()),
                                   C.c_void_p(od.data_ptr() if od is not None else None), 0.5,
                                   C.c_void_p(m.data_ptr()), C.c_void_p(s.data_ptr()), ctx.stream()), 'boxstats')
        ctx.sync()
        mh, sh = m.cpu().numpy(), s.cpu().numpy()
        assert np.array_equal(np.isnan(mh), np.isnan(med_o)) and np.isnan(med_o).any()
        ok = ~np.isnan(med_o)
        assert np.array_equal(mh[ok], med_o[ok])
        np.testing.assert_allclose(sh[ok], std_o[ok], rtol=2e-6, atol=1e-7)
        assert sh[0, 0] == 0 and mh[0, 0] == F(17.25)


def test_box_statistics_bracket_path_equals_full_sort_and_oracle(ctx):
    """bbx_bkg_boxstats takes the clipped statistics from a sorted bracket around the median + the list of wing pixels and
    leaves boxes where that does not hold to the full sort (BBX_OPT_BKG_FULL_SORT = 1: all of them).  Continuous sky with
    stars, cosmic-ray-like outliers on both sides, masked pixels, a gradient across a box, a box that is half star, a box with
    a few hundred usable pixels: the same medians bit for bit from both paths and from the oracle, std within 2e-6."""
    from blackbox_amd._lib import lib, check
    import ctypes as C
    rs = np.random.RandomState(7)
    box, nby, nbx = 60, 10, 12
    ny, nx = nby * box, nbx * box
    yy, xx = np.mgrid[0:ny, 0:nx]
    data = 300 + 0.3 * xx + 12 * rs.standard_normal((ny, nx))
    for _ in range(400):                                            # stars of all sizes
        y0, x0, f, w = rs.uniform(0, ny), rs.uniform(0, nx), 10 ** rs.uniform(2, 5), rs.uniform(1.2, 4)
        r2 = (yy - y0) ** 2 + (xx - x0) ** 2
        sel = r2 < (8 * w) ** 2
        data[sel] += f * np.exp(-r2[sel] / (2 * w * w))
    data[rs.random_sample((ny, nx)) < 0.003] -= 400                # negative outliers
    data[0:box, 0:box] += 40 * (xx[0:box, 0:box] > 30)             # a step inside one box: two populations
    data[box:2 * box, 0:box] += 2000 * np.exp(-((yy[box:2 * box, 0:box] - 90.) ** 2 + (xx[box:2 * box, 0:box] - 30.) ** 2) / 800.)
    data = data.astype(F)
    mask = np.zeros((ny, nx), np.uint8)
    mask[rs.random_sample((ny, nx)) < 0.03] = 1
    mask[2 * box:3 * box, 0:box][rs.random_sample((box, box)) < 0.45] = 4          # just above limfrac
    objmask = (rs.random_sample((ny, nx)) < 0.02).astype(np.uint8)
    med_o, std_o = Z.get_back_mini(data, mask, objmask, box=box)
    t_data, t_mask, t_obj = dev(ctx, data), dev(ctx, mask), dev(ctx, objmask)
    out = {}
    try:
        for full in (0, 1):
            check(lib.bbx_set_option(ctx.h, 8, full), 'bbx_set_option', ctx.h)
            m = torch.full((nby, nbx), -1.0, dtype=torch.float32, device=ctx.device)
            s = torch.full((nby, nbx), -1.0, dtype=torch.float32, device=ctx.device)
            check(lib.bbx_bkg_boxstats(ctx.h, ny, nx, box, C.c_void_p(t_data.data_ptr()), C.c_void_p(t_mask.data_ptr()),
                                       C.c_void_p(t_obj.data_ptr()), 0.5, C.c_void_p(m.data_ptr()), C.c_void_p(s.data_ptr()),
                                       ctx.stream()), 'boxstats')
            ctx.sync()
            out[full] = (m.cpu().numpy(), s.cpu().numpy())
    finally:
        check(lib.bbx_set_option(ctx.h, 8, 0), 'bbx_set_option', ctx.h)
    ok = ~np.isnan(med_o)
    assert ok.sum() >= nby * nbx - 2
    for full in (0, 1):
        mh, sh = out[full]
        assert np.array_equal(np.isnan(mh), np.isnan(med_o))
        assert np.array_equal(mh[ok], med_o[ok]), 'medians, full_sort=%d' % full
        np.testing.assert_allclose(sh[ok], std_o[ok], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(out[0][1][ok], out[1][1][ok], rtol=3e-7, atol=0)


def test_background_mesh(ctx):
    rs = np.random.RandomState(1)
    box, nby, nbx = 20, 12, 16
    ny, nx = nby * box, nbx * box
    yy, xx = np.mgrid[0:ny, 0:nx]
    data = (250 + 0.05 * xx + 0.02 * yy + rs.normal(0, 15, (ny, nx))).astype(F)
    data[rs.random_sample((ny, nx)) < 0.01] += 3000          # stars / CRs the clip must reject
    mask = np.zeros((ny, nx), np.uint8)
    mask[40:100, 60:140] = 4                                  # fully masked boxes -> NaN -> filled
    mask[rs.random_sample((ny, nx)) < 0.05] |= 1
    data[200:210, 10:30] = 0                                  # zeros are excluded (mask_value = 0)
    objmask = (rs.random_sample((ny, nx)) < 0.03).astype(np.uint8)
    med_o, std_o = Z.get_back_mini(data, mask, objmask, box=box)
    assert np.isnan(med_o).sum() >= 6
    d_med, d_std = torch.empty(0), None
    # raw box statistics
    from blackbox_amd._lib import lib, check
    import ctypes as C
    m = torch.empty((nby, nbx), dtype=torch.float32, device=ctx.device)
    s = torch.empty((nby, nbx), dtype=torch.float32, device=ctx.device)
    t_data, t_mask, t_obj = dev(ctx, data), dev(ctx, mask), dev(ctx, objmask)      # keep alive across the launch
    check(lib.bbx_bkg_boxstats(ctx.h, ny, nx, box, C.c_void_p(t_data.data_ptr()), C.c_void_p(t_mask.data_ptr()),
                               C.c_void_p(t_obj.data_ptr()), 0.5, C.c_void_p(m.data_ptr()), C.c_void_p(s.data_ptr()),
                               ctx.stream()), 'boxstats')
    ctx.sync()
    mh, sh = m.cpu().numpy(), s.cpu().numpy()
    assert np.array_equal(np.isnan(mh), np.isnan(med_o))
    ok = ~np.isnan(med_o)
    # medians are order statistics of float32 data: exact; std: float64 sums in another order
    assert np.array_equal(mh[ok], med_o[ok])
    np.testing.assert_allclose(sh[ok], std_o[ok], rtol=2e-6)
    # fill + filter, then the whole get_back
    med_f = Z.fill_filter_mini(med_o)
    gm, gs = G.get_back(ctx, dev(ctx, data), dev(ctx, mask), dev(ctx, objmask), bkg_boxsize=box)
    ctx.sync()
    assert np.array_equal(gm.cpu().numpy(), med_f)
    np.testing.assert_allclose(gs.cpu().numpy(), Z.fill_filter_mini(std_o), rtol=2e-6)
    # mini2back across the frame, fused with the subtraction
    bkg_o = Z.mini2back(med_f, (ny, nx), box)
    d_data = dev(ctx, data)
    bkg = G.mini2back(ctx, gm, (ny, nx), bkg_boxsize=box, subtract_from=d_data)
    ctx.sync()
    # float64 16-tap sums rounded to float32: 1 ulp
    np.testing.assert_allclose(bkg.cpu().numpy(), bkg_o, rtol=2.4e-7)
    np.testing.assert_allclose(d_data.cpu().numpy(), data - bkg_o, rtol=0, atol=1e-4)
    # per-channel zoom (interp_Xchan=False): 2 x 8 channel blocks
    std_f = Z.fill_filter_mini(std_o)
    b2 = G.mini2back(ctx, std_f, (ny, nx), bkg_boxsize=box, interp_Xchan=False)
    ctx.sync()
    np.testing.assert_allclose(b2.cpu().numpy(), Z.mini2back(std_f, (ny, nx), box, channels=(nby // 2, nbx // 8)), rtol=2.4e-7)


def gauss_psf(L, fwhm, dx=0.0, dy=0.0, half=12):
    """Moffat(beta=2.5) stamp of (2*half+1)^2 pixels, like a PSFEx model image, centred on
    pixel [0,0].  (An untruncated analytic Gaussian has a spectrum that falls to 1e-9, where
    any single-precision FFT -- rocFFT or FFTW alike -- only carries rounding noise and
    1/sqrt(den) amplifies it; real PSF models have a finite support and a noise floor.)"""
    a = fwhm / (2 * np.sqrt(2 ** (1 / 2.5) - 1))
    y = np.fft.fftfreq(L) * L
    yy, xx = np.meshgrid(y, y, indexing='ij')
    p = (1 + ((yy - dy) ** 2 + (xx - dx) ** 2) / (a * a)) ** -2.5
    p[(np.abs(yy) > half) | (np.abs(xx) > half)] = 0
    return (p / p.sum()).astype(F)


def make_subs(L, nsub, seed, same=False, inject=None):
    rs = np.random.RandomState(seed)
    N, Rr, Pn, Pr, Vn, Vr, sc = [], [], [], [], [], [], []
    for k in range(nsub):
        sn, sr = 12.0 + k, 6.0 + 0.5 * k
        pn, pr = gauss_psf(L, 4.0 + 0.3 * k), gauss_psf(L, 3.2)
        truth = np.zeros((L, L))
        for _ in range(6):                                   # static stars, present in both
            y, x, f = rs.randint(8, L - 8), rs.randint(8, L - 8), rs.uniform(2e3, 2e4)
            truth[y, x] += f
        conv = lambda img, p: np.fft.ifft2(np.fft.fft2(img) * np.fft.fft2(p.astype(np.float64))).real
        fn, fr = 1.0, 0.8 + 0.1 * k
        new = fn * conv(truth, pn)
        ref = fr * conv(truth, pr)
        if inject is not None:
            t = np.zeros((L, L)); t[inject[0], inject[1]] = inject[2]
            new = new + fn * conv(t, pn)
        if same:
            ref, pr, sr, fr = new.copy(), pn.copy(), sn, fn
            noise_n = noise_r = rs.normal(0, sn, (L, L))
        else:
            noise_n, noise_r = rs.normal(0, sn, (L, L)), rs.normal(0, sr, (L, L))
        N.append(new + noise_n); Rr.append(ref + noise_r); Pn.append(pn); Pr.append(pr)
        Vn.append(np.full((L, L), sn * sn) + np.maximum(new, 0)); Vr.append(np.full((L, L), sr * sr) + np.maximum(ref, 0))
        sc.append([sn, sr, fn, fr, 0.05, 0.04])
    f = lambda a: np.stack(a).astype(F)
    return f(N), f(Rr), f(Pn), f(Pr), f(Vn), f(Vr), np.array(sc, F)


@pytest.mark.parametrize('L', [64, 100, 75])
def test_zogy_vs_oracle(ctx, L):
    N, Rr, Pn, Pr, Vn, Vr, sc = make_subs(L, 3, L)
    outs = G.run_zogy(ctx, *[dev(ctx, a) for a in (N, Rr, Pn, Pr, Vn, Vr)], sc)
    ctx.sync()
    outs = [o.cpu().numpy() for o in outs]
    for k in range(3):
        ref = Z.run_zogy(N[k], Rr[k], Pn[k], Pr[k], sc[k, 0], sc[k, 1], sc[k, 2], sc[k, 3], Vn[k], Vr[k], sc[k, 4], sc[k, 5])
        for name, a, b in zip(('D', 'S', 'Scorr', 'Fpsf', 'Fpsferr'), [o[k] for o in outs], ref):
            scale = np.abs(b).max()
            # single-precision FFT pipelines of different factorisation: 2e-4 of the image scale
            assert np.abs(a - b).max() <= 2e-4 * scale, (name, k, np.abs(a - b).max(), scale)


def test_zogy_properties(ctx):
    L = 128
    # identical new and ref (same noise realisation, same PSF): D vanishes
    N, Rr, Pn, Pr, Vn, Vr, sc = make_subs(L, 2, 7, same=True)
    D, S, Scorr, Fpsf, Fpsferr = [o.cpu().numpy() for o in G.run_zogy(ctx, *[dev(ctx, a) for a in (N, Rr, Pn, Pr, Vn, Vr)], sc)]
    assert np.abs(D).max() < 1e-2 * np.abs(N).max() * 1e-2
    # pure noise: Scorr ~ N(0, 1) (QC ranges Z-SCMED 0 +- 0.3, Z-SCSTD 1 +- 0.15, set_qc.py:382-383)
    rs = np.random.RandomState(3)
    sn, sr = 10.0, 5.0
    N = rs.normal(0, sn, (1, L, L)).astype(F); Rr = rs.normal(0, sr, (1, L, L)).astype(F)
    Pn = gauss_psf(L, 4.0)[None]; Pr = gauss_psf(L, 3.0)[None]
    Vn = np.full((1, L, L), sn * sn, F); Vr = np.full((1, L, L), sr * sr, F)
    sc = np.array([[sn, sr, 1.0, 1.0, 0.0, 0.0]], F)
    D, S, Scorr, Fpsf, Fpsferr = [o.cpu().numpy() for o in G.run_zogy(ctx, *[dev(ctx, a) for a in (N, Rr, Pn, Pr, Vn, Vr)], sc)]
    assert abs(np.median(Scorr)) < 0.3 and abs(Scorr.std() - 1.0) < 0.15
    # injected point source of known flux in the new image: Fpsf at the position = flux
    N, Rr, Pn, Pr, Vn, Vr, sc = make_subs(L, 1, 11, inject=(60, 70, 5.0e4))
    D, S, Scorr, Fpsf, Fpsferr = [o.cpu().numpy() for o in G.run_zogy(ctx, *[dev(ctx, a) for a in (N, Rr, Pn, Pr, Vn, Vr)], sc)]
    assert Fpsf[0, 60, 70] == pytest.approx(5.0e4, rel=0.05)
    assert Scorr[0, 60, 70] > 20 and np.unravel_index(np.argmax(Scorr[0]), (L, L)) == (60, 70)


def test_cut_stitch_and_photometry(ctx):
    rs = np.random.RandomState(2)
    ny, nx, size, border = 96, 144, 48, 5
    img = rs.normal(0, 1, (ny, nx)).astype(F)
    subs = G.cut_subimages(ctx, dev(ctx, img), size, border)
    ctx.sync()
    assert np.array_equal(subs.cpu().numpy(), Z.cut_subimages(img, size, border))
    back = G.stitch_subimages(ctx, subs, (ny, nx), size, border)
    ctx.sync()
    assert np.array_equal(back.cpu().numpy(), img)
    # optimal flux on stamps, incl. sources at the frame edge and pixels with V <= 0
    S, nsrc = 9, 40
    V = np.abs(rs.normal(100, 10, (ny, nx))).astype(F)
    V[10:14, 20:24] = 0
    ys = rs.randint(0, ny, nsrc); xs = rs.randint(0, nx, nsrc)
    ys[:3] = [0, ny - 1, 12]; xs[:3] = [0, nx - 1, 22]
    psfs = np.abs(rs.normal(0, 1, (nsrc, S, S))).astype(F)
    psfs /= psfs.sum(axis=(1, 2), keepdims=True)
    fo, eo = Z.psf_optflux(img, V, psfs, ys, xs)
    fg, eg = G.psf_optflux(ctx, dev(ctx, img), dev(ctx, V), dev(ctx, psfs), ys, xs)
    ctx.sync()
    np.testing.assert_allclose(fg.cpu().numpy(), fo, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(eg.cpu().numpy(), eo, rtol=1e-5)


def test_psf_optflux_sigma_vs_oracle(ctx):
    """the entry the operator's catalogue photometry runs (bbx_psf_optflux_sigma, zogy.get_psfoptflux with the
    variance formed on the fly): D = background-subtracted frame, sigma image -> V = max(D, 0) + sigma^2 in float32
    at the stamp pixels only; against the oracle's psf_optflux on that variance image.  Sources at the frame
    edges and corners, negative pixels, a patch of zero variance (D <= 0 and sigma = 0), one stamp per source."""
    rs = np.random.RandomState(7)
    ny, nx, S, nsrc = 120, 176, 11, 300
    D = rs.normal(0, 12, (ny, nx)).astype(F)
    for _ in range(30):
        y, x = rs.randint(3, ny - 3), rs.randint(3, nx - 3)
        D[y - 2:y + 3, x - 2:x + 3] += rs.uniform(50, 5000)
    sig = np.abs(rs.normal(12, 1.5, (ny, nx))).astype(F)
    sig[40:46, 60:66] = 0
    D[40:46, 60:66] = -np.abs(D[40:46, 60:66])                 # V = 0 there: skipped by both sides
    V = (np.maximum(D, F(0)) + sig * sig).astype(F)
    ys = rs.randint(0, ny, nsrc); xs = rs.randint(0, nx, nsrc)
    ys[:6] = [0, ny - 1, 0, ny - 1, 43, 2]; xs[:6] = [0, nx - 1, nx - 1, 0, 63, nx - 2]
    psfs = np.abs(rs.normal(0, 1, (nsrc, S, S))).astype(F)
    psfs /= psfs.sum(axis=(1, 2), keepdims=True)
    fo, eo = Z.psf_optflux(D, V, psfs, ys, xs)
    fg, eg = G.psf_optflux(ctx, dev(ctx, D), dev(ctx, sig), dev(ctx, psfs), ys, xs, v_is_sigma=True)
    ctx.sync()
    # float64 sums in wave order, rounded to float32 (same bar as bbx_psf_optflux)
    np.testing.assert_allclose(fg.cpu().numpy(), fo, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(eg.cpu().numpy(), eo, rtol=1e-5)
    # and the explicit-variance entry on the same variance image gives the same numbers
    f2, e2 = G.psf_optflux(ctx, dev(ctx, D), dev(ctx, V), dev(ctx, psfs), ys, xs)
    ctx.sync()
    assert torch.equal(f2, fg) and torch.equal(e2, eg)


def test_psf_model_mfma(ctx):
    """a17 PSFEx model evaluation: basis cube x polynomial terms on the f32 MFMA against the
    float32 fma-chain oracle (exact but for rare double roundings) and a float64 contraction"""
    import zogy_core as Z
    from blackbox_amd import zogy as G
    rs = np.random.RandomState(2)
    S, poldeg = 25, 2
    ncoef = (poldeg + 1) * (poldeg + 2) // 2
    yy, xx = np.mgrid[0:S, 0:S] - S // 2
    base = (1 + (yy * yy + xx * xx) / 5.0) ** -2.5
    basis = np.stack([base * (0.3 ** k) * (1 + 0.2 * rs.normal(size=base.shape)) for k in range(ncoef)]).astype(np.float32)
    for nsrc in (1, 37, 1000):
        x, y = rs.uniform(0, 10560, nsrc), rs.uniform(0, 10560, nsrc)
        terms = G.psf_poly_terms(x, y, (5280.0, 5280.0), (5280.0, 5280.0), poldeg)
        assert terms.shape == (nsrc, ncoef)
        got = G.psf_model_stamps(ctx, torch.from_numpy(basis).to(ctx.device), x, y, (5280.0, 5280.0), (5280.0, 5280.0),
                                 poldeg, normalize=False).cpu().numpy().reshape(nsrc, -1)
        want = Z.psf_model(terms, basis.reshape(ncoef, -1))
        ulp = np.spacing(np.abs(want).astype(np.float32))
        assert np.all(np.abs(got - want) <= ulp)
        assert (got != want).mean() < 1e-4
        ref64 = terms.astype(np.float64) @ basis.reshape(ncoef, -1).astype(np.float64)
        np.testing.assert_allclose(got, ref64, rtol=0, atol=2e-7 * np.abs(terms).sum(1, keepdims=True).max() * np.abs(basis).max())
    stamps = G.psf_model_stamps(ctx, torch.from_numpy(basis).to(ctx.device), x, y, (5280.0, 5280.0), (5280.0, 5280.0), poldeg)
    np.testing.assert_allclose(stamps.sum(dim=(1, 2)).cpu().numpy(), 1.0, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('shape,channels', [((176, 176), None), ((176, 176), (88, 22)), ((6, 16), (3, 2)), ((40, 33), None)])
def test_device_spline_prefilter_bits(shape, channels):
    """bbx_spline_prefilter against scipy.ndimage.spline_filter on the edge-padded blocks (zoom_coefficients: what
    scipy.ndimage.zoom(order=3, mode='nearest') filters before it interpolates): the same float64 bits"""
    from blackbox_amd import reduce as R, zogy as G
    ctx = R.Context(0)
    rs = np.random.RandomState(shape[0] * 7 + shape[1])
    mini = (rs.normal(300.0, 20.0, shape) + 50 * np.sin(np.arange(shape[1]) / 7.0)).astype(np.float32)
    mini[0, 0] = 0.0
    ref = G.zoom_coefficients(mini, channels)
    got = G.device_zoom_coefficients(ctx, torch.from_numpy(mini).to(ctx.device), channels).cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), np.abs(got - ref).max()
    ctx.close()


@pytest.mark.gpu
def test_catalogue_candidates_listed_by_the_zoom_equal_the_search_pass():
    """bbx_mini_median == np.median (float32 rule, odd and even counts, NaN); bbx_zoom_candidates: the pixels above
    (float)(median x nsigma) listed by the kernel that writes the background-subtracted frame give the same peaks as
    bbx_find_peaks' own pass; a threshold that is not that number fails the search instead of answering wrongly"""
    from blackbox_amd import reduce as R, zogy as G
    from blackbox_amd._lib import lib, check, BBXError
    ctx = R.Context(0)
    dev_ = ctx.device
    rs = np.random.RandomState(21)
    for n in (1, 2, 7, 1024, 1025, 30976, 30977, 32768, 32769, 100001):
        a = rs.normal(8 if n % 3 else -8, 2, n).astype(np.float32)
        if n == 1024:
            a[:] = np.round(a)                                         # ties across the middle
        out = torch.empty(1, dtype=torch.float32, device=dev_)
        check(lib.bbx_mini_median(ctx.h, n, G._p(torch.from_numpy(a).to(dev_)), G._p(out), ctx.stream()), 'bbx_mini_median')
        assert out.item() == np.median(a), n
    a[5] = np.nan
    check(lib.bbx_mini_median(ctx.h, a.size, G._p(torch.from_numpy(a).to(dev_)), G._p(out), ctx.stream()), 'bbx_mini_median')
    assert np.isnan(out.item())
    box, ny, nx = 20, 480, 640
    img = rs.normal(100, 5, (ny, nx)).astype(np.float32)
    for _ in range(60):
        y, x = rs.randint(3, ny - 3), rs.randint(3, nx - 3)
        img[y - 1:y + 2, x - 1:x + 2] += rs.uniform(30, 3000)
    img[100:140, 200:260] += 500.0                                   # a block of hits: the workgroup's queue overflows into direct appends
    d_img = torch.from_numpy(img).to(dev_)
    mask = torch.zeros((ny, nx), dtype=torch.uint8, device=dev_)
    mini, mstd = G.get_back(ctx, d_img, mask, bkg_boxsize=box)
    nsig = 5.0
    d_med = torch.empty(1, dtype=torch.float32, device=dev_)
    check(lib.bbx_mini_median(ctx.h, mstd.numel(), G._p(mstd), G._p(d_med), ctx.stream()), 'bbx_mini_median')
    thr = float(nsig) * float(np.median(mstd.cpu().numpy()))
    assert np.float32(thr) == np.float32(float(d_med.item()) * nsig)
    work = torch.empty_like(d_img)
    check(lib.bbx_zoom_candidates(ctx.h, G._p(d_med), nsig), 'bbx_zoom_candidates')
    G.mini2back(ctx, mini, (ny, nx), bkg_boxsize=box, subtract_from=d_img, subtract_into=work)
    a = G.find_peaks_arrays(ctx, work, thr, max_out=20000)           # from the list
    b = G.find_peaks_arrays(ctx, work.clone(), thr, max_out=20000)   # own pass
    c = G.find_peaks_arrays(ctx, work, thr, max_out=20000)           # the list is spent: own pass
    assert a[0].size > 60
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    # another threshold on a listed frame: refused
    check(lib.bbx_zoom_candidates(ctx.h, G._p(d_med), nsig), 'bbx_zoom_candidates')
    G.mini2back(ctx, mini, (ny, nx), bkg_boxsize=box, subtract_from=d_img, subtract_into=work)
    with pytest.raises(BBXError):
        G.find_peaks_arrays(ctx, work, thr * 1.5, max_out=20000)
    ctx.close()
