"""a12 back end (acstools.satdet.make_mask as sat_detect calls it, blackbox.py:4197-4213): the oracle restatement -- and,
with a GPU, the HIP path -- against fixtures made with the library functions the absent package consists of
(oracle/gen_golden_sat.py mask -> tests/golden/sat_mask.npz: scikit-image 0.18.3 rotate, numpy median, astropy 4.3.1
sigma_clipped_stats / biweight_midvariance), and the segment handed to it against what scikit-image's probabilistic
Hough transform (acstools' line finder, random by construction) returns for ten seeds."""
import hashlib
import json
import os

import numpy as np
import pytest

import sattrail as S
from blackbox_amd import synth

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sat_mask.npz'))
META = json.loads(str(G['meta']))


def scene(name):
    p = synth.SAT_MASK_SCENES[name]
    img, truth = synth.sat_full_scene(p['seed'], p['ny'], p['nx'], p['trail'])
    assert hashlib.sha256(img.tobytes()).hexdigest() == META[name]['sha_input']
    return img, truth


def golden_mask(name, shape):
    return np.unpackbits(G[name + '_mask'])[:shape[0] * shape[1]].reshape(shape).astype(bool)


def test_library_primitives_restated_exactly():
    """numpy's pairwise sum, astropy's clipped mean and biweight midvariance on short vectors: checked through the
    fixtures' recorded values below; here the pieces that have closed forms"""
    rs = np.random.RandomState(0)
    for n in (3, 7, 8, 9, 10, 16, 23):
        a = rs.normal(0, 1, n)
        assert S._np_sum(a) == np.sum(a)
    x = rs.uniform(0, 2, 50)
    np.testing.assert_allclose(S._pow4(x), x ** 4, rtol=4e-16)
    m, osh = S.rotate_matrix((300, 450), float(np.cos(np.radians(21.1))), float(np.sin(np.radians(21.1))))
    assert osh == (442, 528) and m[2].tolist() == [0.0, 0.0, 1.0]
    u, v = S.to_output(m, 17.0, 230.0)
    assert abs(m[0, 0] * u + m[0, 1] * v + m[0, 2] - 17.0) < 1e-11 and abs(m[1, 0] * u + m[1, 1] * v + m[1, 2] - 230.0) < 1e-11


@pytest.mark.parametrize('name', sorted(META))
def test_oracle_make_mask_vs_library_version(name):
    m = META[name]
    img, truth = scene(name)
    mask_full, nsats, info = S.detect(img)
    assert nsats == 1 and info['segment'] == m['segment']
    b = S.bin2(img)
    mask, dbg = S.make_mask(b, m['segment'], return_debug=True)
    assert np.array_equal(mask, golden_mask(name, b.shape))            # the mask: every pixel
    assert m['same_with_float32_rotation']                             # (scikit-image 0.18's float32 interpolation: same mask)
    w = dbg['windows']
    assert len(w) == m['nwin'] and [list(x['box']) for x in w] == m['boxes'] and [x['z'] for x in w] == m['z']
    assert dbg['deg'] == pytest.approx(m['deg'], rel=1e-14) and list(dbg['rot_shape']) == m['rot_shape']
    # rotation + medians + statistics: to rounding (bit for bit when numpy's cos / matmul agree with the ones of the
    # environment the fixture was made in; the last bits of the rotation matrix depend on that)
    np.testing.assert_allclose(np.concatenate([x['medarr'] for x in w]), G[name + '_medarr'], rtol=1e-12, atol=0)
    np.testing.assert_allclose(np.array([x['mean'] for x in w]), G[name + '_mean'], rtol=1e-12)
    np.testing.assert_allclose(np.array([x['var'] for x in w]), G[name + '_var'], rtol=1e-9)
    # what BlackBOX does with it: 2 x 2 un-binning, one 8-connected trail, the trail covered
    assert np.array_equal(mask_full.astype(bool), np.kron(mask, np.ones((2, 2), bool)))
    # (the walk starts on an EDGE pixel of the trail -- the segment's first point -- so its first windows may hold only
    # the near half of a wide trail until the centre has followed it: acstools' behaviour, 'upright' shows it)
    assert (mask_full.astype(bool) & truth).sum() >= (0.85 if name == 'upright' else 0.95) * truth.sum()
    assert (mask_full.astype(bool) & ~truth).sum() <= 0.02 * truth.size


@pytest.mark.parametrize('name', sorted(META))
def test_segment_inside_probabilistic_hough_envelope(name):
    """the deterministic segment against the segments skimage.transform.probabilistic_hough_line returned for seeds 0..9
    (threshold 210, line_length 200, line_gap 75, acstools' theta grid): its direction lies within their range
    (+- 0.15 deg: their end points are whole pixels of >= 200 px segments), its end points within 2.5 px of one of their lines"""
    seg = np.array(META[name]['segment'], float)
    pht = G[name + '_pht'].astype(float)
    ang = np.degrees(np.arctan2(pht[:, 3] - pht[:, 1], pht[:, 2] - pht[:, 0])) % 180.0
    mine = np.degrees(np.arctan2(seg[1, 1] - seg[0, 1], seg[1, 0] - seg[0, 0])) % 180.0
    # (angles near 0 / 180 wrap: compare on the circle)
    d = (ang - mine + 90.0) % 180.0 - 90.0
    assert d.min() - 0.15 <= 0.0 <= d.max() + 0.15, (mine, sorted(ang))
    best = np.inf
    for x0, y0, x1, y1, _ in pht:
        n = np.hypot(x1 - x0, y1 - y0)
        dist = [abs((px - x0) * (y1 - y0) - (py - y0) * (x1 - x0)) / n for px, py in seg]
        best = min(best, max(dist))
    assert best <= 2.5, best


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(META))
def test_hip_make_mask_vs_library_version(name):
    """bbx_sat_trails on the fixture scenes: the mask bit 16 equals the un-binned golden mask (made with scikit-image's
    rotate and astropy's statistics) pixel for pixel, one trail counted"""
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from blackbox_amd import reduce as R
    img, truth = scene(name)
    ctx = R.Context(0)
    try:
        mask0 = np.zeros(img.shape, np.uint8)
        mask0[::5, ::9] = 1
        d_mask = torch.from_numpy(mask0.copy()).to(ctx.device)
        d_n, d_info = R.sat_detect(ctx, torch.from_numpy(img).to(ctx.device), {}, d_mask, {})
        ctx.sync()
        got = d_mask.cpu().numpy()
        want = np.kron(golden_mask(name, (img.shape[0] // 2, img.shape[1] // 2)), np.ones((2, 2), bool))
        assert np.array_equal((got & 16) != 0, want)
        assert np.array_equal(got & ~np.uint8(16), mask0)                  # the other bits untouched
        info = d_info.cpu().numpy()
        assert int(d_n.item()) == 1 and int(info[5]) == META[name]['nwin'] - sum(1 for z in META[name]['z'] if not z)
    finally:
        ctx.close()
