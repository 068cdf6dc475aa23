"""row f3 (reference co-add) on the GPU: bbx_coadd_prep / bbx_resample_lanczos3 / bbx_coadd_combine
against the CPU restatement (oracle/coadd.py)."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

import coadd as OC                                   # noqa: E402  oracle/coadd.py
from blackbox_amd import coadd as PC                 # noqa: E402
from blackbox_amd import reduce as R                 # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    c = R.Context(0)
    yield c
    c.close()


def dev(ctx, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)


def test_prep_bit_exact(ctx):
    rs = np.random.RandomState(11)
    ny, nx = 330, 517                                            # odd size: exercises the scalar tail
    data = rs.normal(100, 10, (ny, nx)).astype('float32')
    bkg = rs.normal(90, 1, (ny, nx)).astype('float32')
    bstd = np.abs(rs.normal(8, 1, (ny, nx))).astype('float32')
    bstd[rs.random_sample((ny, nx)) < 0.01] = 0
    mask = np.zeros((ny, nx), np.uint8)
    mask[rs.random_sample((ny, nx)) < 0.05] = 4
    mask[rs.random_sample((ny, nx)) < 0.05] |= 1
    mask[rs.random_sample((ny, nx)) < 0.02] |= 16
    mask[:4] = 32; mask[10, 10] = 33
    for discard, nimg, with_bkg in ((49, 3, True), (63, 2, False), (49, 1, True)):
        d = dev(ctx, data)
        _, w = PC.prep_inputimage(ctx, d, dev(ctx, bkg) if with_bkg else None, dev(ctx, bstd), dev(ctx, mask),
                                  masktype_discard=discard, nimages=nimg)
        ctx.sync()
        assert np.array_equal(d.cpu().numpy(), OC.prep_data(data, bkg if with_bkg else None, mask, 32))
        assert np.array_equal(w.cpu().numpy(), OC.prep_weights(bstd, mask, discard, nimg))


def test_scale_chan_zps(ctx):
    rs = np.random.RandomState(12)
    ysz, xsz = 40, 24
    geom = R.geometry((2 * (ysz + 20), 8 * (xsz + 180)), ysz, xsz)
    data = rs.normal(100, 10, (2 * ysz, 8 * xsz)).astype('float32')
    zpc = [None if c == 5 else 22.0 + 0.01 * c for c in range(16)]
    h = {'PC-ZP': 22.1}
    h.update({'PC-ZP{}'.format(c + 1): v for c, v in enumerate(zpc) if v is not None})
    d = dev(ctx, data)
    PC.scale_chan_zps(ctx, d, h, geom)
    ctx.sync()
    assert np.array_equal(d.cpu().numpy(), OC.scale_chan_zps(data, 22.1, zpc, ysz, xsz))
    assert h['PC-ZP3'] == 22.1 and 'PC-ZP6' not in h


def _scene(rs, ny, nx):
    y, x = np.mgrid[0:ny, 0:nx]
    img = rs.normal(0, 1, (ny, nx))
    for _ in range(25):
        cx, cy, a = rs.uniform(5, nx - 5), rs.uniform(5, ny - 5), rs.uniform(50, 2000)
        img += a * np.exp(-0.5 * ((x - cx) ** 2 + (y - cy) ** 2) / 1.6 ** 2)
    return img.astype('float32')


def test_resample_vs_oracle(ctx):
    """rotation + scale + shift through a coarse lattice; zero-weight pixels and the frame border"""
    rs = np.random.RandomState(13)
    ny, nx = 200, 260
    img = _scene(rs, ny, nx)
    w = rs.uniform(0.01, 0.03, (ny, nx)).astype('float32')
    w[rs.random_sample((ny, nx)) < 0.002] = 0
    th = np.deg2rad(3.0)

    def f(yy, xx):
        return (130 + 1.02 * ((xx - 120) * np.cos(th) - (yy - 90) * np.sin(th)) + 0.3,
                100 + 1.02 * ((xx - 120) * np.sin(th) + (yy - 90) * np.cos(th)) - 0.45)
    # the third map has 0.97 input px per output px (fits the staged box), the fourth 1.6 (a 64-pixel
    # tile row spans 102 input pixels, more than the stage holds): the gather path runs
    out_shape, step = (190, 250), 32
    for fun, fscale in ((f, 1.0), (f, 1.7), (lambda yy, xx: (3.0 + 2.55 * xx * 0.38, 4.0 + 2.55 * yy * 0.38), 1.0),
                        (lambda yy, xx: (2.5 + 1.6 * xx + 0.05 * yy, 3.5 - 0.04 * xx + 1.6 * yy), 1.0)):
        grid = OC.coarse_grid(fun, out_shape[0], out_shape[1], step)
        xin, yin = OC.grid_positions(grid, out_shape[0], out_shape[1], step)
        o_ref, w_ref = OC.lanczos3_resample(img, w, xin, yin, fscale)
        o, wo = PC.resample(ctx, dev(ctx, img), dev(ctx, w), grid, out_shape, fscale, step)
        ctx.sync()
        o, wo = o.cpu().numpy(), wo.cpu().numpy()
        assert (w_ref == 0).sum() > 500 and (w_ref > 0).sum() > 10000
        assert np.array_equal(wo == 0, w_ref == 0)                 # same footprint decisions
        # float32 accumulation in the same order; the taps may differ in their last bit
        # (device sin vs numpy sin): 2e-6 of the image scale
        assert np.abs(o - o_ref).max() <= 2e-6 * np.abs(o_ref).max()
        ok = w_ref > 0
        assert np.abs(wo[ok] / w_ref[ok] - 1).max() < 2e-5


def test_resample_rejects_short_grid(ctx):
    img = dev(ctx, np.zeros((64, 64), 'float32'))
    grid = np.zeros((2, 3, 2))                                    # covers 32 x 64 output pixels at step 32
    with pytest.raises(ValueError):
        PC.resample(ctx, img, img, grid, (40, 40), 1.0, 32)
    with pytest.raises(ValueError):
        PC.resample(ctx, img, img, grid, (32, 65), 1.0, 32)
    PC.resample(ctx, img, img, grid, (32, 64), 1.0, 32)
    ctx.sync()


@pytest.mark.parametrize('n', [3, 7, 12, 20])
def test_combine_vs_oracle(ctx, n):
    rs = np.random.RandomState(14 + n)
    ny, nx = 96, 131
    cube = rs.normal(10, 1, (n, ny, nx)).astype('float32')
    wc = rs.uniform(0.5, 2, (n, ny, nx)).astype('float32')
    wc[rs.random_sample(wc.shape) < 0.1] = 0
    wc[:, 7, 7] = 0
    wc[1:, 8, 8] = 0                                              # a single valid value
    cube[rs.random_sample(cube.shape) < 0.01] += 300.0            # outliers
    cube[0, 20, :40] = cube[1, 20, :40]                           # ties
    dc, dw = dev(ctx, cube), dev(ctx, wc)
    sig, amp = float(np.float32(3.5)), float(np.float32(0.3))
    for t in ('weighted', 'average', 'median', 'clipped', 'min', 'max', 'sum'):
        o_ref, w_ref, nclip_ref = OC.combine(cube, wc, t, clip_sigma=sig, clip_ampfrac=amp)
        o, wo, nclip, cm = PC.combine(ctx, dc, dw, t, nsigma_clip=sig, A_swarp=amp, clipmask=True)
        ctx.sync()
        assert np.array_equal(o.cpu().numpy(), o_ref), t            # float64 sums in the same order
        assert np.array_equal(wo.cpu().numpy(), w_ref), t
        if t == 'clipped':
            assert np.array_equal(nclip.cpu().numpy(), nclip_ref)
            assert nclip_ref.sum() > 0
            nsig_ref, drop_ref = OC.clip_nsigma(cube, wc, sig, amp)
            assert np.array_equal(cm[0].cpu().numpy() != 0, drop_ref)
            assert np.array_equal(cm[1].cpu().numpy(), nsig_ref)
            assert np.array_equal(cm[0].cpu().numpy().reshape(n, -1).sum(axis=1), nclip_ref)
        else:
            assert cm is None and int(nclip.sum()) == 0
    with pytest.raises(ValueError):
        PC.combine(ctx, dc, dw, 'mode')


def test_imcombine_dithered_clipped(ctx):
    """three dithered exposures of one sky, a cosmic ray in one of them: the clipped co-add equals
    the oracle's and does not contain the cosmic ray"""
    rs = np.random.RandomState(21)
    ny, nx = 220, 240
    sky_wcs = PC.TanWCS([200.0, -40.0], [120.5, 110.5], [[-1.56e-4, 0], [0, 1.56e-4]])
    big = _scene(rs, 300, 320)                                    # the "true" sky on a larger frame
    wbig = np.ones_like(big)
    true_wcs = PC.TanWCS([200.0, -40.0], [160.5, 150.5], [[-1.56e-4, 0], [0, 1.56e-4]])
    images, weights, wcss, pos = [], [], [], []
    for k, (dra, ddec, rot) in enumerate(((0.0, 0.0, 0.0), (0.0011, -0.0007, 0.4), (-0.0009, 0.0013, -0.3))):
        th = np.deg2rad(rot)
        cd = 1.56e-4 * np.array([[-np.cos(th), np.sin(th)], [np.sin(th), np.cos(th)]])
        wk = PC.TanWCS([200.0 + dra, -40.0 + ddec], [120.5, 110.5], cd)
        g = PC.projection_grid(true_wcs, wk, (ny, nx), 32)        # exposure k = the sky seen through its WCS
        xin, yin = OC.grid_positions(g, ny, nx, 32)
        img, _ = OC.lanczos3_resample(big, wbig, xin, yin)
        img = (img + rs.normal(0, 1, img.shape)).astype('float32')
        if k == 1:
            img[100:104, 90] += 800.0                             # cosmic ray
        images.append(img); weights.append(np.full(img.shape, 1.0, 'float32')); wcss.append(wk)
        gk = PC.projection_grid(wk, sky_wcs, (ny, nx), 32)
        pos.append(OC.grid_positions(gk, ny, nx, 32))
    fs = [1.0, 1.1, 0.95]
    sig, amp = 4.0, float(np.float32(0.3))
    ref, wref, nclip_ref = OC.coadd(images, weights, pos, fs, 'clipped', clip_sigma=sig, clip_ampfrac=amp)
    refw, _, _ = OC.coadd(images, weights, pos, fs, 'weighted')
    out, wout, nclip, _ = PC.imcombine(ctx, [dev(ctx, a) for a in images], [dev(ctx, a) for a in weights], wcss, sky_wcs,
                                       (ny, nx), 'clipped', fs, sig, amp)
    ctx.sync()
    out, wout = out.cpu().numpy(), wout.cpu().numpy()
    scale = np.abs(ref).max()
    # resampled planes agree to ~2e-6 of the scale; a pixel sitting on a clip threshold may flip
    differ = np.abs(out - ref) > 1e-5 * scale
    assert differ.mean() < 1e-4
    assert abs(int(nclip.sum()) - int(nclip_ref.sum())) <= 3
    cr = refw[95:110, 85:96].max() - ref[95:110, 85:96].max()
    assert cr > 100                                               # the weighted mean keeps the CR, the clipped one does not
    assert out[95:110, 85:96].max() < ref[95:110, 85:96].max() + 1e-3 * scale
    inner = (slice(20, 200), slice(20, 220))
    assert (wout[inner] > 0).all()
    # the reference's two passes: the clip log goes back to the input frames (clipped2mask), the
    # weights of the filtered pixels are zeroed, the second pass is a plain weighted mean
    dws = [dev(ctx, a) for a in weights]
    masks = [dev(ctx, np.zeros(a.shape, np.uint8)) for a in images]
    out2, wout2, nclip2, _ = PC.imcombine(ctx, [dev(ctx, a) for a in images], dws, wcss, sky_wcs, (ny, nx), 'clipped', fs,
                                          sig, amp, masks=masks, fwhm=[3.0, 3.0, 3.0])
    ctx.sync()
    out2 = out2.cpu().numpy()
    w1 = dws[1].cpu().numpy()
    assert (w1[100:104, 90] == 0).all() and (w1 == 0).sum() < 200          # the cosmic ray, little else
    assert (dws[0].cpu().numpy() == 0).sum() < 100
    assert out2[95:110, 85:96].max() < ref[95:110, 85:96].max() + 0.05 * scale   # gone from the second pass too
    assert np.isfinite(out2).all()


def test_clipped2mask_vs_oracle(ctx):
    """the clip log carried to an input frame, filtered like pass_filters, saturated neighbourhoods
    released, weights zeroed: integer work, identical to the numpy restatement"""
    rs = np.random.RandomState(31)
    out_shape, in_shape, step = (150, 180), (140, 170), 32
    th = np.deg2rad(2.0)

    def f(yy, xx):
        return (3.0 + 0.97 * (xx * np.cos(th) - yy * np.sin(th)), 6.0 + 0.97 * (xx * np.sin(th) + yy * np.cos(th)) - 4.0)
    grid = OC.coarse_grid(f, out_shape[0], out_shape[1], step)
    xin, yin = OC.grid_positions(grid, out_shape[0], out_shape[1], step)
    clip = rs.random_sample(out_shape) < 0.004
    nsig = (rs.normal(0, 3.0, out_shape) + np.where(rs.random_sample(out_shape) < 0.5, 4.5, -4.5)).astype('float32')
    # a satellite-trail like streak and a blob: clusters that the 5x5 box filter picks up
    for t in range(60):
        clip[40 + t // 3, 20 + t] = True; nsig[40 + t // 3, 20 + t] = 3.2
    clip[90:96, 100:107] = True; nsig[90:96, 100:107] = -3.5
    nsig[~clip] = 0
    data_mask = np.zeros(in_shape, np.uint8)
    data_mask[60:64, 60:64] = 4; data_mask[59:65, 59:65] |= 8          # a saturated star near part of the streak
    data_mask[rs.random_sample(in_shape) < 0.01] |= 1
    weights = rs.uniform(0.5, 2.0, in_shape).astype('float32')
    for nsigma_clip, fwhm in ((3.0, 2.0), (2.5, 4.0)):
        m_ref, w_ref = OC.clipped2mask(clip, nsig, xin, yin, in_shape, data_mask, weights, nsigma_clip, fwhm)
        dw = dev(ctx, weights)
        m, nm = PC.clipped2mask(ctx, dev(ctx, clip.astype(np.uint8)), dev(ctx, nsig), grid, in_shape, dev(ctx, data_mask), dw,
                                nsigma_clip, fwhm, step)
        ctx.sync()
        assert m_ref.sum() > 50
        assert np.array_equal(m.cpu().numpy() != 0, m_ref)
        assert np.array_equal(dw.cpu().numpy(), w_ref)
        assert int(nm.item()) == int(m_ref.sum())
