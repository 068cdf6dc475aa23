"""GPU: satellite-trail masking against the deterministic detector of oracle/sattrail.py
(parity with the reference is unpinnable: probabilistic Hough / CNN), plus the synthetic-
line pins of SURVEY.md section 8c: a trail of width w and S/N k across the frame gets
>= 95 % of its pixels masked with < 0.1 % false area; a frame without a trail stays clean."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
if not torch.cuda.is_available():
    pytest.skip('no GPU', allow_module_level=True)

import sattrail as S                        # noqa: E402
from blackbox_amd import reduce as R       # noqa: E402

F = np.float32


@pytest.fixture(scope='module')
def ctx():
    c = R.Context(0)
    yield c
    c.close()


def scene(seed, ny=600, nx=900, trail=None, nstars=150):
    rs = np.random.RandomState(seed)
    img = 250 + rs.normal(0, 18, (ny, nx))
    yy, xx = np.mgrid[0:ny, 0:nx]
    for _ in range(nstars):
        y0, x0, f = rs.uniform(0, ny), rs.uniform(0, nx), 10 ** rs.uniform(3, 5.5)
        r2 = (yy - y0) ** 2 + (xx - x0) ** 2
        sel = r2 < 15 ** 2
        img[sel] += (f / (2 * np.pi * 1.7 ** 2)) * np.exp(-r2[sel] / (2 * 1.7 ** 2))
    truth = np.zeros((ny, nx), bool)
    if trail is not None:
        (xa, ya, xb, yb, amp, width) = trail
        norm = np.hypot(xb - xa, yb - ya)
        d = ((xx - xa) * (yb - ya) - (yy - ya) * (xb - xa)) / norm
        prof = amp * np.exp(-0.5 * (d / (width / 2.355)) ** 2)
        img += prof
        truth = np.abs(d) <= width / 2
    return img.astype(F), truth


@pytest.mark.parametrize('seed,trail', [(1, (0, 120, 900, 470, 120.0, 6.0)), (2, (200, 0, 520, 600, 80.0, 6.0)),
                                        (3, None), (4, (0, 300, 900, 305, 200.0, 8.0))])
def test_sat_trails_vs_oracle(ctx, seed, trail):
    img, truth = scene(seed, trail=trail)
    mask0 = np.zeros(img.shape, np.uint8)
    mask0[::7, ::11] = 1
    m_o, nsats_o, info_o = S.sat_detect(img, mask0.copy())
    d_mask = torch.from_numpy(mask0.copy()).to(ctx.device)
    d_n, d_info = R.sat_detect(ctx, torch.from_numpy(img).to(ctx.device), {}, d_mask, {})
    ctx.sync()
    info = d_info.cpu().numpy()
    assert info[0] == np.float32(info_o['bmax']) and info[1] == np.float32(info_o['bmin'])
    assert int(info[2]) == info_o['votes']
    assert np.array_equal(d_mask.cpu().numpy(), m_o)                 # uint8 mask bit-exact
    assert int(d_n.item()) == nsats_o
    got = (d_mask.cpu().numpy() & 16) != 0
    if trail is None:
        assert nsats_o == 0 and not got.any()
    else:
        assert nsats_o == 1
        assert (got & truth).sum() >= 0.95 * truth.sum()
        assert (got & ~truth).sum() <= 0.02 * truth.size           # strip a few px wider than the FWHM, no stray area
        assert (got & ~ndimage_dilate(truth, 12)).sum() == 0


def ndimage_dilate(m, r):
    from scipy import ndimage
    return ndimage.binary_dilation(m, iterations=r)
