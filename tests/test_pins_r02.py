"""Reference-run fixtures of round 2 (tests/golden/pins_r02.*, made by oracle/gen_golden_r02.py from
the reference's OWN master_prep, get_flatstats, nonlin_corr, mask_init(imgtype='flat') and
qc_check / run_qc_check with the real Settings/set_qc.py): the oracle restatement and -- on a GPU --
the HIP path reproduce them.  PINNED rows: a7 nonlin_corr, a8 master_prep, a14 get_flatstats, f4 QC."""
import json
import os

import numpy as np
import pytest

import bbx_oracle as O
from blackbox_amd import qc, settings, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
META = json.load(open(os.path.join(GOLD, 'pins_r02.json')))
NPZ = np.load(os.path.join(GOLD, 'pins_r02.npz'))
TEL = 'ML1'


def gpu():
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from blackbox_amd import reduce as R
    return torch, R


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---- a8 master frames ------------------------------------------------------------------------
def test_inputs_regenerate_identically():
    for imgtype in ('flat', 'bias'):
        frames, _ = synth.master_frames(imgtype)
        assert [sha(f) for f in frames] == META['master_' + imgtype]['sha_inputs']
    d, m = synth.flatstat_frame()
    assert sha(d) == META['flatstats']['sha_data'] and sha(m) == META['flatstats']['sha_mask']
    assert sha(synth.nonlin_frame()) == META['nonlin']['sha_in']


def test_oracle_master_flat_vs_reference():
    frames, medsec = synth.master_frames('flat')
    master = O.master_median(np.stack(frames), 'flat', medsec=medsec, bpm=synth.master_bpm())
    assert sha(master.astype(np.float32)) == META['master_flat']['sha']
    assert np.array_equal(master[::4], NPZ['master_flat'])
    ys, xs = synth.MASTER_GEOM
    f = O.gain_correction_factors(master, ys, xs)
    want = [META['master_flat']['header']['GAINCF%d' % (c + 1)] for c in range(16)]
    np.testing.assert_allclose(f, want, rtol=3e-16 * 4)       # the FITS card holds 16 significant digits


def test_oracle_master_bias_vs_reference():
    frames, _ = synth.master_frames('bias')
    master = O.master_median(np.stack(frames), 'bias')
    assert sha(master.astype(np.float32)) == META['master_bias']['sha']
    h = META['master_bias']['header']
    # the reference environment (astropy 4.3 + bottleneck) sums the float32 survivors in float32: its own
    # values carry up to ~3e-5 of rounding (the same as RDN{c} in DESIGN.md section 2); the restatement and the product accumulate in float64
    mean, _, std, _ = O.sigma_clipped_stats_median(master)
    assert mean == pytest.approx(h['MBMEAN'], abs=1e-4) and std == pytest.approx(h['MBRDN'], rel=1e-4)
    ys, xs = synth.MASTER_GEOM
    for c in range(16):
        iy, ix = divmod(c, 8)
        mean, _, std, _ = O.sigma_clipped_stats_median(master[iy * ys:(iy + 1) * ys, ix * xs:(ix + 1) * xs])
        assert mean == pytest.approx(h['MBIASM%d' % (c + 1)], abs=1e-4)
        assert std == pytest.approx(h['MBRDN%d' % (c + 1)], rel=1e-4)


@pytest.mark.gpu
def test_gpu_masters_vs_reference():
    torch, R = gpu()
    from blackbox_amd import masters
    ctx = R.Context(0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)      # noqa: E731
    ys, xs = synth.MASTER_GEOM
    frames, medsec = synth.master_frames('flat')
    master = masters.master_median(ctx, [dev(f) for f in frames], 'flat', medsec=medsec, bpm=dev(synth.master_bpm()))
    ctx.sync()
    assert sha(master.cpu().numpy()) == META['master_flat']['sha']                # every pixel of the reference's master
    h = {}
    masters.gain_correction_factors(ctx, master, h, ysize_chan=ys, xsize_chan=xs)
    for c in range(16):
        assert R.hval(h, 'GAINCF%d' % (c + 1)) == pytest.approx(META['master_flat']['header']['GAINCF%d' % (c + 1)], rel=2e-15), c
    frames, _ = synth.master_frames('bias')
    mb = masters.master_median(ctx, [dev(f) for f in frames], 'bias')
    ctx.sync()
    assert sha(mb.cpu().numpy()) == META['master_bias']['sha']
    h = {}
    masters.master_level_stats(ctx, mb, h, 'bias', ysize_chan=ys, xsize_chan=xs)
    want = META['master_bias']['header']
    for k in ['MBMEAN', 'MBRDN'] + ['MBIASM%d' % (c + 1) for c in range(16)] + ['MBRDN%d' % (c + 1) for c in range(16)]:
        # float64 moments here, float32 running sums in the reference environment (see the oracle test)
        assert R.hval(h, k) == pytest.approx(want[k], rel=1e-4, abs=1e-4), k
    ctx.close()


# ---- a14 get_flatstats ---------------------------------------------------------------------------
FLAT_EXACT = ['MEDSEC', 'FLATMED'] + ['FLATM%d' % (c + 1) for c in range(16)]
FLAT_CLOSE = ['STDSEC', 'FLATSTD', 'RSTDSEC', 'FLATRSTD', 'RDIF-MAX', 'RSTD-MAX'] + ['FLATS%d' % (c + 1) for c in range(16)] + \
             ['FLATRS%d' % (c + 1) for c in range(16)]


def test_oracle_flatstats_vs_reference():
    data, mask = synth.flatstat_frame()
    ys, xs = synth.FLATSTAT_GEOM
    got = O.get_flatstats(data, mask, synth.FLATSTAT_SEC, ys, xs, synth.FLATSTAT_SUB)
    want = META['flatstats']['header']
    for k in FLAT_EXACT:
        assert float(got[k]) == want[k], k
    for k in ('STDSEC', 'FLATSTD', 'RDIF-MAX', 'RSTD-MAX'):
        assert float(got[k]) == pytest.approx(want[k], rel=1e-6), k


@pytest.mark.gpu
def test_gpu_flatstats_vs_reference():
    torch, R = gpu()
    from blackbox_amd import flatstats as F
    ctx = R.Context(0)
    data, mask = synth.flatstat_frame()
    ys, xs = synth.FLATSTAT_GEOM
    h = F.get_flatstats(ctx, torch.from_numpy(data).to(ctx.device), {}, torch.from_numpy(mask).to(ctx.device), TEL,
                        statsec=synth.FLATSTAT_SEC, subsize=synth.FLATSTAT_SUB, ysize_chan=ys, xsize_chan=xs)
    want = META['flatstats']['header']
    for k in FLAT_EXACT:
        assert float(R.hval(h, k)) == want[k], k                  # order statistics: exact
    for k in FLAT_CLOSE:
        assert float(R.hval(h, k)) == pytest.approx(want[k], rel=3e-6), k
    assert R.hval(h, 'NSUBS') == want['NSUBS'] and R.hval(h, 'NSUBSTOT') == want['NSUBSTOT'] and R.hval(h, 'STATSEC') == want['STATSEC']
    ctx.close()


# ---- a7 nonlin_corr ----------------------------------------------------------------------------------
def _splines():
    from scipy import interpolate
    return [interpolate.UnivariateSpline(x, y, w=w, k=k, s=s) for (x, y, w, k, s) in synth.nonlin_splines()]


def test_oracle_nonlin_vs_reference():
    spl = _splines()
    # the pickled splines of the reference run have the same knots and coefficients
    for s, (t, c, k) in zip(spl, META['nonlin']['tck']):
        assert np.allclose(s._eval_args[0], t, rtol=0, atol=0) and np.allclose(s._eval_args[1], c, rtol=1e-12) and s._eval_args[2] == k
    ys, xs = synth.NONLIN_GEOM
    data = synth.nonlin_frame()
    out = O.nonlin_corr(data.copy(), spl, settings.gain[TEL], ys, xs)
    assert np.array_equal(out, NPZ['nonlin_out'])
    assert (out[data / np.float32(2.2) > 50000] * 2 == data[data / np.float32(2.2) > 50000]).all()      # the reference's quirk, as run


@pytest.mark.gpu
def test_gpu_nonlin_vs_reference():
    torch, R = gpu()
    ctx = R.Context(0)
    ys, xs = synth.NONLIN_GEOM
    # the reference run's own (t, c, k)
    tck = [(np.array(t), np.array(c), k) for (t, c, k) in META['nonlin']['tck']]
    d = torch.from_numpy(synth.nonlin_frame()).to(ctx.device)
    geom = R.geometry((2 * ys + 40, 8 * xs + 80), ys, xs)
    R.nonlin_corr(ctx, d, geom, TEL, splines=tck)
    ctx.sync()
    assert np.array_equal(d.cpu().numpy(), NPZ['nonlin_out'])
    ctx.close()


# ---- mask_init(imgtype='flat') -----------------------------------------------------------------------
def test_reference_flat_mask_is_the_bpm():
    m = META['mask_flat']
    assert m['equals_bpm'] and m['sha_mask'] == m['sha_bpm'] and not m['saturate_in_header'] and m['header_mask'] == {}


# ---- f4 QC ---------------------------------------------------------------------------------------------
def test_qc_check_vs_reference():
    """the product's qc_check / run_qc_check on the same headers as the reference's (real set_qc ranges):
    same flagged keywords, colours, QC-FLAG and QC{RED,ORA,YEL}n keywords"""
    cases = {c[0]: c for c in synth.qc_headers()}
    assert len(META['qc']) == len(cases)
    for want in META['qc']:
        name, tel, ktype, hd = cases[want['name']]
        h = dict(hd)
        keys, colors, ranges, comments = qc.qc_check(h, telescope=tel, check_key_type=ktype, return_range_comment=True)
        assert list(keys) == want['keys'] and list(colors) == want['colors'], name
        assert [str(r) for r in ranges] == want['ranges'], name
        h2 = {k: (v, '') for k, v in hd.items()}                  # (value, comment) cards: the comments are kept
        assert qc.run_qc_check(h2, tel, check_key_type=ktype) == want['flag'], name
        added = {k: (v[0] if isinstance(v, tuple) else v) for k, v in h2.items() if k not in hd}
        assert added == want['added'], name
        for k, c in want['added_comments'].items():
            assert (h2[k][1] if isinstance(h2[k], tuple) else '') == c, (name, k)


# ---- f3: clip log -> input-frame masks (in-reference integer code) ------------------------------------
def test_oracle_pass_filters_vs_reference():
    """oracle/coadd.pass_filters == the reference's own buildref.pass_filters on the same table; the HIP
    kernel (bbx_clipped2mask) is held to that oracle in tests/test_gpu_coadd.py"""
    import coadd as OC
    x, y, ns, shape = synth.clip_points()
    assert META['pass_filters']['npoints'] == x.size and tuple(META['pass_filters']['shape']) == shape
    for name, (fsize, fsigma, fmax) in synth.CLIP_FILTERS.items():
        m = OC.pass_filters(x, y, ns, list(fsize), list(fsigma), list(fmax), shape)
        want = np.unpackbits(NPZ['passfilt_' + name])[:shape[0] * shape[1]].reshape(shape).astype(bool)
        assert META['pass_filters']['cases'][name]['n'] == int(want.sum()) > 0
        assert np.array_equal(m, want), name


# ---- f4: verify_header ------------------------------------------------------------------------------------
def test_verify_header_vs_reference_behaviour():
    """for every keyword of the product's contract that the reference's verify_header knows ('full'
    headers; per-channel families: channels 1 and 16 there), the same reaction to the keyword missing
    (KeyError / warning), to 'None' (ValueError / accepted) and to values of each basic type (dtype warning)"""
    probe = META['verify_header']['probe']
    known = {k: e for k, e in probe.items() if e['known']}
    assert len(known) >= 40 and 'RDNOISE' in known and 'COSMIC-P' in known
    sample = {'bool': True, 'int': 3, 'float': 2.5, 'str': 'x'}
    base = {}
    for k, e in qc.REDUCTION_CONTRACT.items():
        base[k] = sample[e['dtype'].__name__]
    assert qc.verify_header(dict(base), ['full']) == []

    def outcome(h, key):
        try:
            w = qc.verify_header(h, ['full'])
        except KeyError:
            return 'KeyError', []
        except ValueError:
            return 'ValueError', []
        return 'ok', [m for m in w if ('keyword ' + key + ' ') in m or ('keyword ' + key + ':') in m]
    for key, e in known.items():
        h = dict(base); del h[key]
        res, w = outcome(h, key)
        assert (res if res != 'ok' else ('warning' if w else 'silent')) == e['missing'], key
        h = dict(base); h[key] = 'None'
        assert outcome(h, key)[0] == e['none'], key
        good = []
        for tname, val in sample.items():
            h = dict(base); h[key] = val
            res, w = outcome(h, key)
            if not any('dtype of keyword' in m for m in w):
                good.append(tname)
        assert good == e['dtype_ok'], key
    # the families the reference spells out for channels 1 and 16 only carry the same rule for every channel
    for fam in ('GAIN', 'BIASM', 'RDN', 'VFITOK'):
        for c in range(2, 16):
            assert qc.REDUCTION_CONTRACT['%s%d' % (fam, c)] == qc.REDUCTION_CONTRACT['%s1' % fam]
