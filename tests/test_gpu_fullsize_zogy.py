"""GPU, BASELINE full size (10560 x 10560, 64 sub-images of L = 1400 = 2^3 * 5^2 * 7 -- the
reference's production sub-image size): background mesh, satellite trail and the ZOGY
subtraction on the whole frame against the oracle on boxes / whole sub-images of it.

Tolerances of the ZOGY images are relative to the LOCAL noise of each image (not to its
maximum): a float32 FFT of a 1400^2 sub-image that holds 10^6 e- stars next to a 20 e- sky
noise has rounding errors of ~1e-7 of the bright pixels everywhere; both sides (HIP and the
numpy complex64 oracle) carry them."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
if not torch.cuda.is_available():
    pytest.skip('no GPU', allow_module_level=True)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench                                   # noqa: E402  (synthetic full-size frame generator)
import sattrail as S                           # noqa: E402
import zogy_core as Z                          # noqa: E402
from blackbox_amd import reduce as R           # noqa: E402
from blackbox_amd import zogy as G             # noqa: E402

F = np.float32
YSZ, XSZ, OS_Y, OS_X = 5280, 1320, 20, 180
NY, NX = 2 * YSZ, 8 * XSZ
SIZE, BORDER, BOX = 1320, 40, 60
L = SIZE + 2 * BORDER
TRAIL = (0.0, 2100.0, float(NX), 6400.0, 90.0, 6.0)


@pytest.fixture(scope='module')
def scene():
    ctx = R.Context(0)
    raw, flat, bpm, ex = bench.synth_frame_device(torch, ctx.device, YSZ, XSZ, OS_Y, OS_X, 3000, 'u16', extras=True,
                                                  ntrans=40, trail=TRAIL)
    rs = np.random.RandomState(0)
    coeffs = np.zeros((16, 16)); coeffs[~np.eye(16, dtype=bool)] = rs.uniform(0, 2e-4, 240)
    stages = {}
    data, mask, header, hm = R.reduce_object(ctx, raw, {}, 'ML1', mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0,
                                             stages=stages)
    del raw, flat
    ref, ref_mask = bench.synth_reference(torch, ctx.device, ex['scene0'], 3000)
    del ex['scene0']
    torch.cuda.empty_cache()
    sc = dict(ctx=ctx, data=data, mask=mask, header=header, hm=hm, ref=ref, ref_mask=ref_mask, trans=ex['transients'],
              pre_sat=stages['data_xtalk'], bpm=bpm)
    yield sc
    ctx.close()


def oracle_mesh(scene, which='new'):
    """the ORACLE's background mesh of the reduced frame (or of the reference with its own sky): filled + filtered mini
    images, the background-subtracted frame and the sigma image, all made by oracle/zogy_core.py from the pixels -- nothing
    of the product's subtraction stage enters (kept with the module's scene: ~1 min of numpy per frame)"""
    key = 'oracle_mesh_' + which
    if key not in scene:
        hd = (scene['data'] if which == 'new' else scene["ref"] + 120.0).cpu().numpy()
        hm = (scene['mask'] if which == 'new' else scene['ref_mask']).cpu().numpy()
        med_raw, std_raw = Z.get_back_mini(hd, hm, None, box=BOX)
        med_o, std_o = Z.fill_filter_mini(med_raw), Z.fill_filter_mini(std_raw)
        sub_o = (hd - Z.mini2back(med_o, (NY, NX), BOX)).astype(F)
        sig_o = Z.mini2back(std_o, (NY, NX), BOX, channels=(med_o.shape[0] // 2, med_o.shape[1] // 8) if which == 'new' else None)
        scene[key] = dict(med_raw=med_raw, med=med_o, std=std_o, sub=sub_o, sig=sig_o)
    return scene[key]


def test_reduce_flags(scene):
    h = scene['header']
    for k in ('GAIN-P', 'OS-P', 'MFLAT-P', 'MASK-P', 'COSMIC-P', 'XTALK-P', 'SAT-P'):
        assert R.hval(h, k) is True, k
    assert R.hval(h, 'NSATS') == 1


def test_sat_trail_fullsize(scene):
    """the full frame through bbx_sat_trails == the oracle detector on the full frame (mask
    bit-exact, level / sigma / votes equal); the injected trail is covered"""
    ctx = scene['ctx']
    # the frame as it was when sat_detect ran (after crosstalk, before edge fill) and the mask
    # without the trail bit
    pre = scene['pre_sat']
    m0 = (scene['mask'] & ~16)
    d_mask = m0.clone()
    d_n, d_info = R.sat_detect(ctx, pre, {}, d_mask, {})
    ctx.sync()
    info = d_info.cpu().numpy()
    m_o, nsats_o, info_o = S.sat_detect(pre.cpu().numpy(), m0.cpu().numpy())
    assert info[0] == np.float32(info_o['bmax']) and info[1] == np.float32(info_o['bmin'])
    assert int(info[2]) == info_o['votes']
    assert int(d_n.item()) == nsats_o == 1
    got = d_mask.cpu().numpy()
    assert np.array_equal(got, m_o)
    assert np.array_equal(got, scene['mask'].cpu().numpy())      # what reduce_object left
    xa, ya, xb, yb, amp, width = TRAIL
    yy, xx = np.mgrid[0:NY:7, 0:NX:7]
    d = ((xx - xa) * (yb - ya) - (yy - ya) * (xb - xa)) / np.hypot(xb - xa, yb - ya)
    truth = np.abs(d) <= width / 2
    hit = (got[0:NY:7, 0:NX:7] & 16) != 0
    assert (hit & truth).sum() >= 0.95 * truth.sum()
    assert (hit & (np.abs(d) > 30)).sum() == 0


def test_background_mesh_fullsize(scene):
    """get_back / mini2back on the full frame: every box median equal to the oracle's, std to
    2e-6, the bicubic zoom (176 x 176 -> 10560 x 10560; per channel for the sigma image) to
    float32 rounding of scipy.ndimage.zoom"""
    ctx = scene['ctx']
    data, mask = scene['data'], scene['mask']
    med, std = G.get_back(ctx, data, mask, bkg_boxsize=BOX)
    ctx.sync()
    om = oracle_mesh(scene)
    assert om['med_raw'].shape == (NY // BOX, NX // BOX) and np.isnan(om['med_raw']).any()      # edge boxes are fully masked
    med_o, std_o = om['med'], om['std']
    mh, sh = med.cpu().numpy(), std.cpu().numpy()
    assert np.array_equal(mh, med_o)
    np.testing.assert_allclose(sh, std_o, rtol=3e-6)
    work = data.clone()
    G.mini2back(ctx, med, (NY, NX), bkg_boxsize=BOX, interp_Xchan=True, subtract_from=work, want_bkg=False)
    bstd = G.mini2back(ctx, std, (NY, NX), bkg_boxsize=BOX, interp_Xchan=False)
    ctx.sync()
    got = work.cpu().numpy()
    # |bkg| ~ 250 e-: one float32 ulp of the background is 3e-5
    assert np.abs(got - om['sub']).max() <= 6.2e-5
    del got
    np.testing.assert_allclose(bstd.cpu().numpy(), om['sig'], rtol=2e-6)


PSF_S = 49


def tile_medians(mini):
    """median of a mini image over the boxes of each sub-image (what zogy hands run_ZOGY as sigma_n, sigma_r)"""
    bs = SIZE // BOX
    nsy, nsx = NY // SIZE, NX // SIZE
    return np.median(mini.reshape(nsy, bs, nsx, bs).transpose(0, 2, 1, 3).reshape(nsy * nsx, bs * bs), axis=1).astype(F)


@pytest.mark.parametrize('branch', ['ref-with-mesh', 'ref-bkgsub'])
def test_zogy_fullsize(scene, branch):
    """optimal_subtraction on the full frame in the SURVEY 8d configuration -- 64 sub-images of 1400^2, one 49 x 49 Moffat PSF
    pair PER SUB-IMAGE (FWHM gradient across the field), flux ratio, sigma_n, sigma_r, dx, dy per sub-image
    (blackbox.py:3754-3759, call 2460-2465) -- against the oracle's run_zogy on whole sub-images (a corner, an interior
    one, the last one).  The oracle's inputs are made BY THE ORACLE from the reduced frame: its own background mesh,
    background-subtracted frames, sigma images, variance images max(x, 0) + sigma^2 and per-sub-image sigma scalars
    (oracle_mesh); nothing of the product's subtraction stage is fed back to it.  Every injected transient is recovered with
    its flux; the transient list of those sub-images and the PSF photometry of the catalogue against the oracle.  Both ways
    a reference comes: with its own sky (mesh + sigma image made here) or as buildref delivers it -- background-subtracted
    with its `_bkg_std_mini` image ('ref-bkgsub': the configuration bench.py times)."""
    ctx = scene['ctx']
    nsy, nsx = NY // SIZE, NX // SIZE
    zi = bench.zogy_inputs(torch, ctx.device, nsy, nsx, PSF_S, BOX, NY, NX)
    pn_all, pr_all = zi['psf_new'].cpu().numpy(), zi['psf_ref'].cpu().numpy()
    assert pn_all.shape == (64, PSF_S, PSF_S) and len({p.tobytes() for p in pn_all}) == 64
    fr_all, dx_all, dy_all = zi['fratio'], zi['dx'], zi['dy']
    om = oracle_mesh(scene)
    if branch == 'ref-bkgsub':
        ref_in = scene['ref']
        kw = dict(ref_is_bkgsub=True, ref_bkg_std_mini=zi['ref_bkg_std_mini'])
        Rw_o = scene['ref'].cpu().numpy()
        sdr_o = zi['ref_bkg_std_mini']
        rsig_o = Z.mini2back(sdr_o, (NY, NX), BOX)
    else:
        ref_in = scene["ref"] + 120.0                                              # a reference that still carries its sky
        kw = dict(ref_is_bkgsub=False)
        omr = oracle_mesh(scene, 'ref')
        Rw_o, sdr_o, rsig_o = omr['sub'], omr['std'], omr['sig']
    res = G.optimal_subtraction(ctx, scene['data'], ref_in, scene['mask'], scene['ref_mask'], zi['psf_new'], zi['psf_ref'],
                                fratio=fr_all, dx=dx_all, dy=dy_all, cat_extract=True, **kw)
    ctx.sync()
    if branch == 'ref-bkgsub':
        assert res['ref_bkgsub'] is scene['ref'] and 'bkg_mini_ref' not in res          # no mesh of the reference
    hdr = res['header_trans']
    assert res['header_new']['Z-P'][0] is True and hdr['Z-SIZE'][0] == SIZE and hdr['Z-BSIZE'][0] == BORDER
    assert 'Z-KWIN' not in hdr                                                          # the row window (228 of 1400 rows at S = 49) held
    # the oracle's side: frames, sigma images, variance images, scalars
    Nw_o, nsig_o = om['sub'], om['sig']
    Vn_o = (np.maximum(Nw_o, F(0)) + nsig_o * nsig_o).astype(F)
    Vr_o = (np.maximum(Rw_o, F(0)) + rsig_o * rsig_o).astype(F)
    sn_o, sr_o = tile_medians(om['std']), tile_medians(np.asarray(sdr_o, F))
    # the scalars the product handed its kernels: sigma_n, sigma_r are medians of (identical / 3e-6-close) mini images
    np.testing.assert_allclose(res['scal'][:, 0], sn_o, rtol=3e-6)
    np.testing.assert_allclose(res['scal'][:, 1], sr_o, rtol=3e-6)
    assert len(set(res['scal'][:, 0].tolist())) > 32 and len(set(res['scal'][:, 1].tolist())) > 8
    Nw = res['data_bkgsub']

    def embed(p):
        k = np.zeros((L, L), F); h = p.shape[0] // 2
        for j in range(p.shape[0]):
            for i in range(p.shape[1]):
                k[(j - h) % L, (i - h) % L] = p[j, i]
        return k

    def cut(t, sy, sx):
        """one padded sub-image from a host frame"""
        out = np.zeros((L, L), F)
        y0, x0 = sy * SIZE - BORDER, sx * SIZE - BORDER
        ya, yb, xa, xb = max(y0, 0), min(y0 + L, NY), max(x0, 0), min(x0 + L, NX)
        out[ya - y0:yb - y0, xa - x0:xb - x0] = t[ya:yb, xa:xb]
        return out
    worst = {}
    trans_h = res['transients']
    for (sy, sx) in ((0, 0), (3, 4), (7, 7)):
        k = sy * nsx + sx
        D, Sm, Sc, Fp, Fe = Z.run_zogy(cut(Nw_o, sy, sx), cut(Rw_o, sy, sx), embed(pn_all[k]), embed(pr_all[k]), sn_o[k], sr_o[k],
                                       1.0, 1.0 / fr_all[k], cut(Vn_o, sy, sx), cut(Vr_o, sy, sx), dx_all[k], dy_all[k])
        inner = (slice(BORDER, BORDER + SIZE), slice(BORDER, BORDER + SIZE))
        tile = (slice(sy * SIZE, (sy + 1) * SIZE), slice(sx * SIZE, (sx + 1) * SIZE))
        # float32 rounding of a 2-D FFT spreads along the row and the column of a bright pixel (the
        # row pass and the column pass each round at ~1e-7 of the largest value in their line): the
        # images carry errors of ~2e-6 of the brightest input pixel of their row / column, on both
        # sides (tools/dbg/zogy_dbg.py: HIP and numpy complex64 are equally far from a float64
        # evaluation, 0.22 / 0.24 e- on the rows of a saturated star with 1.4e5 e- pixels)
        a = np.abs(cut(Nw_o, sy, sx)) + np.abs(cut(Rw_o, sy, sx))
        big = np.maximum(a.max(axis=1)[:, None], a.max(axis=0)[None, :])[inner]
        fe = Fe[inner]
        tols = {}
        for key, want in (('D', D), ('Scorr', Sc), ('Fpsf', Fp), ('Fpsferr', Fe)):
            got = res[key][tile].cpu().numpy()
            want = want[inner]
            # local noise of the image: 1.4826 * MAD of the oracle tile (Fpsferr: its median level)
            noise = np.median(want) if key == 'Fpsferr' else 1.4826 * np.median(np.abs(want - np.median(want)))
            ok = np.isfinite(want)
            assert np.array_equal(np.isfinite(got), ok)
            # per-pixel tolerance: 5e-3 of the local noise + 4e-6 of the brightest input pixel of the row /
            # column, in the image's own units (the matched filter sums ~ N_eff = 36 pixels of D for
            # Fpsf: x 6; Scorr = S / sigma_S with sigma_S / F_S = Fpsferr)
            unit = {'D': 1.0, 'Fpsf': 6.0, 'Fpsferr': 6.0, 'Scorr': 6.0 / np.maximum(fe, 1e-3)}[key]
            tol = 5e-3 * noise + 4e-6 * big * unit
            tols[key] = tol
            err = np.abs(got - want)
            worst[(key, k)] = (float((err[ok] / noise).max()), float((err[ok] / tol[ok]).max()))
            assert (err[ok] <= tol[ok]).all(), (key, (sy, sx), worst[(key, k)], noise)
            # on lines without a bright star (nothing above 2000 e-) the local-noise term alone holds
            sky = ok & (big < 2e3)
            assert sky.sum() > 0.3 * sky.size and (err[sky] <= 6e-3 * noise).all(), (key, (sy, sx), float((err[sky] / noise).max()))
        # ---- the transient list of this sub-image (get_trans: 8-connected regions of |Scorr| >= 6, peak pixel, Fpsf
        # and Fpsferr there) against the oracle's finder on the oracle's own Scorr.  Compared: the regions that lie
        # inside the tile, a pixel off its seams (beyond a seam the stitched frame holds the neighbouring sub-image's
        # evaluation, which agrees with this one's border only roughly -- that is what the borders are for); a peak of
        # the product inside the tile must be such a region's, or belong to an oracle region that leaves the tile.
        # Peaks within 0.05 of the threshold may exist on one side only (the two images differ by the tolerance above).
        y0, x0 = sy * SIZE - BORDER, sx * SIZE - BORDER
        sc_o = np.where(np.isfinite(Sc), Sc, 0).astype(F)
        lab_o, regs = Z.find_transients_fast(sc_o, 6.0, regions=True)
        inside = lambda b: b[0] > BORDER and b[1] < BORDER + SIZE and b[2] > BORDER and b[3] < BORDER + SIZE      # noqa: E731
        want_t = {(y + y0, x + x0): v for (y, x, v, box) in regs if inside(box)}
        open_lab = {k + 1 for k, r in enumerate(regs) if not inside(r[3])}
        got_t = {(t['y'], t['x']): t for t in trans_h
                 if sy * SIZE + 1 <= t['y'] < (sy + 1) * SIZE - 1 and sx * SIZE + 1 <= t['x'] < (sx + 1) * SIZE - 1
                 and int(lab_o[t['y'] - y0, t['x'] - x0]) not in open_lab}
        def nearby(p, pool):
            """the same peak on the other side: the same pixel, or -- when two pixels of a region's top agree
            within the images' tolerance -- its neighbour"""
            return [q for q in pool if abs(q[0] - p[0]) <= 1 and abs(q[1] - p[1]) <= 1]

        sure_w = {p for p, v in want_t.items() if abs(v) >= 6.05}
        sure_g = {p for p, t in got_t.items() if abs(t['scorr']) >= 6.05}
        miss_w = sorted(p for p in sure_w if not nearby(p, got_t))
        miss_g = sorted(p for p in sure_g if not nearby(p, want_t))
        assert not miss_w and not miss_g, ((sy, sx), miss_w, miss_g)
        assert len(sure_w) >= 3
        nsame = 0
        for p in sure_w:
            q = p if p in got_t else nearby(p, got_t)[0]
            nsame += q == p
            t = got_t[q]
            yy, xx = q[0] - y0, q[1] - x0
            # the values the operator reports = the images at the peak: the images' own per-pixel tolerance
            ti = (yy - BORDER, xx - BORDER)
            assert t['scorr'] == pytest.approx(float(Sc[yy, xx]), abs=float(tols['Scorr'][ti])), (q, t)
            assert t['fpsferr'] == pytest.approx(float(Fe[yy, xx]), abs=float(tols['Fpsferr'][ti])), (q, t)
            assert t['fpsf'] == pytest.approx(float(Fp[yy, xx]), abs=float(tols['Fpsf'][ti])), (q, t)
        assert nsame >= 0.9 * len(sure_w)
    print('ZOGY full size, max |HIP - oracle| (/ local noise, / tolerance):', worst)
    # Scorr of the unmasked frame ~ N(0, 1) (QC ranges set_qc.py:382-383)
    assert abs(hdr['Z-SCMED'][0]) < 0.3 and abs(hdr['Z-SCSTD'][0] - 1) < 0.15
    # the header statistics themselves: astropy's sigma_clipped_stats (oracle restatement, pinned by tests/golden/sigclip.npz)
    # of the lattice of every 8th pixel of the HIP frames, pixels with mask bits other than the cosmic-ray flag left out
    import bbx_oracle as O
    sel = (scene['mask'].cpu().numpy()[::8, ::8] & ~np.uint8(2)) == 0
    for key_m, key_s, name in (('Z-SCMED', 'Z-SCSTD', 'Scorr'), ('Z-FPEMED', 'Z-FPESTD', 'Fpsferr')):
        lat = res[name][::8, ::8].cpu().numpy()
        _, med, std, n = O.sigma_clipped_stats_median(np.where(sel, lat, np.nan))
        assert hdr[key_m][0] == float(med), name                                  # exact order statistic
        assert hdr[key_s][0] == pytest.approx(float(std), rel=1e-10), name
    # injected transients: found within a pixel, flux within 3 sigma + 5 %
    found = {(t['y'], t['x']): t for t in res['transients']}
    mask_h = scene['mask'].cpu().numpy()
    nfound = 0
    for (ty, tx, fl) in scene['trans']:
        if mask_h[ty - 3:ty + 4, tx - 3:tx + 4].any():
            continue                                              # on a masked / cleaned spot
        near = [t for (y, x), t in found.items() if abs(y - ty) <= 1 and abs(x - tx) <= 1]
        snr = fl / near[0]['fpsferr'] if near else 0
        if fl < 1500 and not near:
            continue                                              # faint ones may fall below 6 sigma
        assert near, (ty, tx, fl)
        assert abs(near[0]['fpsf'] - fl) <= 3.5 * near[0]['fpsferr'] + 0.06 * fl, (ty, tx, fl, near[0])
        nfound += 1
    assert nfound >= 20
    # ---- catalogue (a17): the source list = peaks of the 8-connected regions of the background-subtracted frame
    # above 5 x S-BKGSTD on unmasked pixels, PSF-weighted optimal flux of each with the variance max(D, 0) + sigma^2
    # (bbx_psf_optflux_sigma) -- positions identical to the oracle's finder on the same frame, fluxes of 1500
    # sources spread over the list against the oracle's psf_optflux
    cat = res['catalog']
    assert cat is not None and len(cat['X_POS']) > 5000
    assert np.isfinite(cat['E_FLUX_OPT']).all() and (cat['E_FLUXERR_OPT'] > 0).all()
    work_h, sig_h = Nw.cpu().numpy(), nsig_o
    thr = 5.0 * res['header_new']['S-BKGSTD'][0]
    peaks_o = [(y, x, v) for (y, x, v) in Z.find_transients_fast(work_h, thr) if v > 0 and mask_h[y, x] == 0]
    ys_c, xs_c = cat['Y_POS'].astype(np.int64) - 1, cat['X_POS'].astype(np.int64) - 1
    assert [(y, x) for y, x, _ in peaks_o] == list(zip(ys_c.tolist(), xs_c.tolist()))
    assert np.array_equal(cat['E_FLUX_PEAK'], np.asarray([v for *_, v in peaks_o], F))
    pick = np.unique(np.concatenate([np.arange(0, ys_c.size, max(1, ys_c.size // 1500)), np.argsort(cat['E_FLUX_PEAK'])[-50:],
                                     np.argsort(ys_c)[:20], np.argsort(xs_c)[-20:]]))
    assert pick.size >= 1000
    V_h = (np.maximum(work_h, F(0)) + sig_h * sig_h).astype(F)
    stamps = pn_all[(ys_c[pick] // SIZE) * nsx + xs_c[pick] // SIZE]                     # the PSF of the sub-image a source falls in
    f_o, e_o = Z.psf_optflux_vec(work_h, V_h, stamps, ys_c[pick], xs_c[pick])
    np.testing.assert_allclose(cat['E_FLUX_OPT'][pick], f_o, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(cat['E_FLUXERR_OPT'][pick], e_o, rtol=1e-5)


def test_zogy_fullsize_scaling_and_null_properties():
    """bbx_zogy_frame on 64 sub-images of 1400^2, properties that hold whatever the frame: (i) new, ref and their noise scalars
    times 4 (a power of two: every operation scales exactly) make Fpsf exactly 4 times larger, bit for bit, and D (in
    units of the new frame's flux) 4 times larger to float32 rounding -- D shares its inverse row transform with V(S), whose
    values do not scale with the frames, so the rounding it picks up there differs; (ii) identical frames with identical PSFs, noise and flux scalars give D = 0 to float32 rounding of the
    transforms and no transient above 6 sigma"""
    import bench
    ctx = R.Context(0)
    dev = ctx.device
    ny = nx = 10560
    g = torch.Generator(device=dev); g.manual_seed(3)
    new = (20 * torch.randn(ny, nx, device=dev, generator=g)).contiguous()
    ref = (8 * torch.randn(ny, nx, device=dev, generator=g)).contiguous()
    new[5000:5003, 7000:7003] += 900.0
    sn = torch.full((ny, nx), 20.0, device=dev); sr = torch.full((ny, nx), 8.0, device=dev)
    psf = torch.from_numpy(np.repeat(bench.moffat_stamp(25, 4.0)[None], 64, 0)).to(dev)
    scal = np.tile(np.array([[20, 8, 1, 0.9, 0.03, 0.02]], np.float32), (64, 1))
    base = G.run_zogy_frame(ctx, new, ref, sn, sr, psf, psf, scal, 1320, 40)
    s4 = scal.copy(); s4[:, 0:2] *= 4
    big = G.run_zogy_frame(ctx, new * 4, ref * 4, sn * 4, sr * 4, psf, psf, s4, 1320, 40)
    ctx.sync()
    assert float((big[0] - base[0] * 4).abs().max()) <= 2e-6 * 4 * float(base[0].abs().max())        # D
    assert bool((big[3] == base[3] * 4).all())                                # Fpsf
    same = np.tile(np.array([[20, 20, 1, 1, 0.0, 0.0]], np.float32), (64, 1))
    null = G.run_zogy_frame(ctx, new, new, sn, sn, psf, psf, same, 1320, 40)
    ctx.sync()
    assert float(null[0].abs().max()) < 1e-4 * float(new.abs().max())        # D: the frames' own scale is 20 / 900
    assert float(null[2].abs().max()) < 1e-3                                  # Scorr
    ctx.close()
