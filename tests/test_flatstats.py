"""a14 get_flatstats (blackbox.py:3661-3820): HIP segment statistics against the oracle's
numpy restatement (deterministic variant), and the oracle's deterministic variant against
the reference's random-subsample estimator."""
import numpy as np
import pytest

import bbx_oracle as O


def flat_frame(seed, ys=192, xs=48):
    rs = np.random.RandomState(seed)
    ny, nx = 2 * ys, 8 * xs
    yy, xx = np.mgrid[0:ny, 0:nx]
    flat = 30000.0 * (1 - 0.1 * ((yy / ny - 0.5) ** 2 + (xx / nx - 0.5) ** 2))
    data = (flat + rs.normal(0, 1, (ny, nx)) * np.sqrt(flat)).astype(np.float32)
    # a few stars and bad pixels
    mask = np.zeros((ny, nx), np.uint8)
    mask[rs.randint(0, ny, 300), rs.randint(0, nx, 300)] = 1
    data[rs.randint(0, ny, 200), rs.randint(0, nx, 200)] += 5e4
    data[5, 7] = np.nan
    return data, mask


def test_oracle_estimator_vs_population():
    data, mask = flat_frame(1)
    sec = (slice(40, 120), slice(100, 300))
    a = O.get_flatstats(data, mask, sec, 192, 48, 96)
    b = O.get_flatstats(data, mask, sec, 192, 48, 96, fraction=0.2, seed=5)
    assert a['MEDSEC'] == b['MEDSEC'] and a['STDSEC'] == b['STDSEC']
    n = 0.2 * data.size
    assert abs(a['FLATMED'] - b['FLATMED']) < 5 * 1.25 * a['FLATSTD'] / np.sqrt(n)
    assert abs(a['FLATSTD'] - b['FLATSTD']) < 0.05 * a['FLATSTD']


@pytest.mark.gpu
def test_gpu_flatstats_vs_oracle():
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from blackbox_amd import reduce as R
    from blackbox_amd import flatstats as F
    ctx = R.Context(0)
    for seed, (ys, xs, sub) in ((1, (192, 48, 96)), (2, (180, 45, 60))):
        data, mask = flat_frame(seed, ys, xs)
        sec = (slice(37, 37 + 75), slice(30, 30 + 301))            # odd sizes: scalar / ragged paths
        want = O.get_flatstats(data, mask, sec, ys, xs, sub)
        h = F.get_flatstats(ctx, torch.from_numpy(data).to(ctx.device), {}, torch.from_numpy(mask).to(ctx.device),
                            'ML1', statsec=sec, subsize=sub, ysize_chan=ys, xsize_chan=xs)
        # medians are order statistics: exact
        for k in ['MEDSEC', 'FLATMED'] + ['FLATM%d' % (c + 1) for c in range(16)]:
            assert R.hval(h, k) == want[k], k
        # sigmas: float64 moments here, float32 pairwise sums in numpy -> 2e-6 relative
        for k in ['STDSEC', 'FLATSTD'] + ['FLATS%d' % (c + 1) for c in range(16)]:
            assert R.hval(h, k) == pytest.approx(want[k], rel=2e-6), k
        assert R.hval(h, 'RDIF-MAX') == pytest.approx(want['RDIF-MAX'], rel=1e-6)
        assert R.hval(h, 'RSTD-MAX') == pytest.approx(want['RSTD-MAX'], rel=2e-6)
        assert R.hval(h, 'RSTDSEC') == pytest.approx(want['STDSEC'] / want['MEDSEC'], rel=2e-6)
        ns = data.shape[0] // sub
        assert R.hval(h, 'NSUBSTOT') == ns * ns
    ctx.close()


@pytest.mark.gpu
def test_gpu_gaincf_vs_oracle():
    """GAINCF{c} (blackbox.py:5085-5161): exact strip medians + float32 channel scalings"""
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from blackbox_amd import reduce as R
    from blackbox_amd import masters
    ctx = R.Context(0)
    ys, xs = 160, 72
    rs = np.random.RandomState(4)
    master = (1.0 + 0.01 * rs.normal(size=(2 * ys, 8 * xs))).astype(np.float32)
    gains = 1 + 0.05 * rs.normal(size=16)
    for c in range(16):
        iy, ix = divmod(c, 8)
        master[iy * ys:(iy + 1) * ys, ix * xs:(ix + 1) * xs] *= np.float32(gains[c])
    want = O.gain_correction_factors(master, ys, xs, nrows_v=20, nrows_h=50, ncols=16)
    h = {}
    got = masters.gain_correction_factors(ctx, torch.from_numpy(master).to(ctx.device), h, ys, xs, nrows_v=20,
                                          nrows_h=50, ncols=16)
    assert np.array_equal(got, want)
    # and they undo the injected gains up to the noise of the strip medians
    rel = want * gains
    assert np.std(rel / rel.mean()) < 3e-3
    assert R.hval(h, 'GAINCF5') == want[4]
    ctx.close()


@pytest.mark.gpu
def test_gpu_master_level_stats():
    """MBMEAN / MBRDN / MBIASM{c} / MBRDN{c}: clipped statistics about the exact median"""
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from blackbox_amd import reduce as R
    from blackbox_amd import masters
    ctx = R.Context(0)
    ys, xs = 120, 90
    rs = np.random.RandomState(8)
    m = rs.normal(0, 3, (2 * ys, 8 * xs)).astype(np.float32)
    for c in range(16):
        iy, ix = divmod(c, 8)
        m[iy * ys:(iy + 1) * ys, ix * xs:(ix + 1) * xs] += np.float32(0.5 * c)
    m[rs.randint(0, 2 * ys, 400), rs.randint(0, 8 * xs, 400)] += 200          # hot pixels: clipped
    m[rs.randint(0, 2 * ys, 300), rs.randint(0, 8 * xs, 300)] = 0              # masked value
    m[3, 3] = np.nan
    h = {}
    full, chan = masters.master_level_stats(ctx, torch.from_numpy(m).to(ctx.device), h, 'bias', ys, xs)
    mean, med, std, n = O.sigma_clipped_stats_median(m)
    assert full[0] == n and np.float32(full[1]) == med
    assert R.hval(h, 'MBMEAN') == pytest.approx(mean, rel=1e-9, abs=1e-9)
    assert R.hval(h, 'MBRDN') == pytest.approx(std, rel=1e-9)
    sec = O.define_sections(m.shape, ys, xs)[4]
    for c in range(16):
        mean, med, std, n = O.sigma_clipped_stats_median(m[sec[c]])
        assert chan[c, 0] == n and np.float32(chan[c, 1]) == med, c
        assert R.hval(h, 'MBIASM%d' % (c + 1)) == pytest.approx(mean, rel=1e-9, abs=1e-9)
        assert R.hval(h, 'MBRDN%d' % (c + 1)) == pytest.approx(std, rel=1e-9)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize('shape,step', [((1003, 1500), 8), ((64, 48), 1), ((700, 900), 3)])
def test_frame_clipped_stats_vs_oracle_and_select_path(shape, step):
    """bbx_frame_clipped_stats (one sort, clipping rounds as index arithmetic) against astropy's algorithm restated
    (oracle.sigma_clipped_stats_median, pinned by tests/golden/sigclip.npz) and against the bracketed-select path
    (bbx_rect_clipped_stats) on the same lattice: n and median exact, mean / std to float64 summation order."""
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from blackbox_amd import reduce as R, zogy as G, flatstats
    ctx = R.Context(0)
    rs = np.random.RandomState(shape[0] + step)
    img = rs.normal(3.0, 2.0, shape).astype(np.float32)
    img[rs.randint(0, shape[0], 2000), rs.randint(0, shape[1], 2000)] += rs.uniform(20, 500, 2000).astype(np.float32)   # outliers: clipped
    img[rs.randint(0, shape[0], 500), rs.randint(0, shape[1], 500)] = 0                # the masked value
    img[::step, ::step][1, 2] = np.nan
    mask = np.zeros(shape, np.uint8)
    mask[rs.randint(0, shape[0], 3000), rs.randint(0, shape[1], 3000)] = rs.choice([1, 2, 4, 16, 32], 3000).astype(np.uint8)
    d_img, d_mask = torch.from_numpy(img).to(ctx.device), torch.from_numpy(mask).to(ctx.device)
    for m, dm in ((None, None), (mask, d_mask)):
        st = G.frame_clipped_stats_enqueue(ctx, d_img, dm, step).cpu().numpy()
        sub = img[::step, ::step].copy()
        if m is not None:
            sub[(m[::step, ::step] & ~np.uint8(2)) != 0] = np.nan                      # bits other than the cosmic-ray flag (2) drop the pixel
        mean, med, std, n = O.sigma_clipped_stats_median(sub)
        assert st[0] == n and np.float32(st[1]) == med
        assert st[2] == pytest.approx(mean, rel=1e-12, abs=1e-12) and st[3] == pytest.approx(std, rel=1e-12)
        dsub = d_img[::step, ::step].contiguous()
        msub = dm[::step, ::step].contiguous() if dm is not None else None
        ref = flatstats.rect_clipped_stats(ctx, dsub, msub, 0, 0, dsub.shape[0], dsub.shape[1], dsub.shape[0], dsub.shape[1], skip_zero=True)[0]
        assert st[0] == ref[0] and st[1] == ref[1]
        assert st[2] == pytest.approx(ref[2], rel=1e-12, abs=1e-12) and st[3] == pytest.approx(ref[3], rel=1e-12)
    # nothing valid: n = 0, NaN median
    st = G.frame_clipped_stats_enqueue(ctx, torch.zeros(16, 16, device=ctx.device), None, 1).cpu().numpy()
    assert st[0] == 0 and np.isnan(st[1])
    ctx.close()
