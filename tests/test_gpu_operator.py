"""GPU: the operator surface beyond the plain reduction -- the failure convention of every step
(`<STEP>-P = False`, carry on; blackbox.py:1476-1952) for host-side and device-side errors, in
the serial path and in the frames-in-flight pipeline; the subtraction products of the CLI
(set_blackbox.py:157-164); --image_list through FramePipeline."""
import importlib.util
import logging
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
if not torch.cuda.is_available():
    pytest.skip('no GPU', allow_module_level=True)

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
import bbx_oracle as O                                  # noqa: E402
from blackbox_amd import _lib, fitsio, settings, synth  # noqa: E402
from blackbox_amd import reduce as R                    # noqa: E402
from blackbox_amd import zogy as G                      # noqa: E402

YS, XS, TEL = 120, 330, 'ML1'


def load_cli():
    spec = importlib.util.spec_from_file_location('bbx_cli', os.path.join(ROOT, 'blackbox.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope='module')
def ctx():
    c = R.Context(0)
    yield c
    c.close()


@pytest.fixture(scope='module')
def case():
    return synth.make_case(YS, XS, 77, tel=TEL, os_y=20, os_x=45, n_stars=60, n_sat=2, n_cr=40)


def dev(ctx, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)


def run(ctx, case, **kw):
    coeffs = O.xtalk_coeffs(case['xtalk'])
    return R.reduce_object(ctx, dev(ctx, case['raw']), {}, TEL, mflat=dev(ctx, case['flat']), bpm=dev(ctx, case['bpm']),
                           xtalk_coeffs=coeffs, exptime=60.0, ysize_chan=YS, xsize_chan=XS, log=logging.getLogger('t'), **kw)


FLAGS = ('GAIN-P', 'OS-P', 'MFLAT-P', 'MASK-P', 'COSMIC-P', 'XTALK-P', 'SAT-P')


def test_all_steps_green(ctx, case):
    d, m, h, hm = run(ctx, case)
    for k in FLAGS:
        assert R.hval(h, k) is True, k


@pytest.mark.parametrize('symbol,flag,key', [('bbx_lacosmic', 'COSMIC-P', 'NCOSMICS'), ('bbx_sat_trails', 'SAT-P', 'NSATS'),
                                            ('bbx_xtalk', 'XTALK-P', None), ('bbx_mask_finish', 'MASK-P', None)])
def test_host_side_failure_of_a_step(ctx, case, monkeypatch, symbol, flag, key):
    """a BBXError raised by one stage's entry point: that stage's flag goes False, every other
    stage still runs and the products come back"""
    good = run(ctx, case)
    monkeypatch.setattr(R.lib, symbol, lambda *a: -2)
    d, m, h, hm = run(ctx, case)
    monkeypatch.undo()
    assert R.hval(h, flag) is False
    for k in FLAGS:
        if k != flag:
            assert R.hval(h, k) is True, k
    if key:
        assert R.hval(h, key) == 'None'
    assert d.shape == good[0].shape and 'M-CRPNUM' in hm
    if symbol == 'bbx_sat_trails':                               # everything before the failed step is unchanged
        assert torch.equal((m & ~16), (good[1] & ~16))


def test_os_corr_failure_adopts_zero_overscan(ctx, case, monkeypatch):
    """blackbox.py:1537-1585: the frame is cropped with an overscan of zero, BIASM = 0, RDN = 10"""
    def boom(*a, **k):
        raise RuntimeError('injected')
    monkeypatch.setattr(R, 'os_solve', boom)
    d, m, h, hm = R.reduce_object(ctx, dev(ctx, case['raw']), {}, TEL, mflat=None, bpm=None, exptime=60.0, ysize_chan=YS,
                                  xsize_chan=XS, do_cosmics=False, detect_sats=False)
    monkeypatch.undo()
    assert R.hval(h, 'OS-P') is False and R.hval(h, 'RDNOISE') == 10.0 and R.hval(h, 'BIASMEAN') == 0.0
    assert R.hval(h, 'BIASM7') == 0.0 and R.hval(h, 'RDN16') == 10.0
    raw = case['raw'].astype(np.float32)
    gain = np.float32(settings.gain[TEL])
    dy, dx = raw.shape[0] // 2, raw.shape[1] // 8
    want = np.empty((2 * YS, 8 * XS), np.float32)
    for c in range(16):
        iy, ix = divmod(c, 8)
        y0 = iy * dy + (0 if iy == 0 else dy - YS)
        want[iy * YS:(iy + 1) * YS, ix * XS:(ix + 1) * XS] = raw[y0:y0 + YS, ix * dx:ix * dx + XS] * gain[c]
    got = d.cpu().numpy()
    edge = (m.cpu().numpy() & 32) != 0
    assert np.array_equal(got[~edge], want[~edge])


def test_device_side_overflow_flags_the_step_only(ctx, case):
    """LA-Cosmic work lists too small (BBX_OPT_DEBUG_LISTCAP): the device raises the overflow flag
    asynchronously; it is attributed to the cosmics step (bbx_step_mark) -> COSMIC-P False,
    NCOSMICS 'None', the later steps ran"""
    _lib.check(_lib.lib.bbx_set_option(ctx.h, 2, 8), 'bbx_set_option')
    try:
        d, m, h, hm = run(ctx, case)
    finally:
        _lib.check(_lib.lib.bbx_set_option(ctx.h, 2, 0), 'bbx_set_option')
    assert R.hval(h, 'COSMIC-P') is False and R.hval(h, 'NCOSMICS') == 'None'
    for k in ('OS-P', 'MASK-P', 'XTALK-P', 'SAT-P'):
        assert R.hval(h, k) is True, k
    # the context is clean again: the next frame is fine
    d2, m2, h2, _ = run(ctx, case)
    assert R.hval(h2, 'COSMIC-P') is True


def test_pipeline_flags_one_frame_and_continues(ctx, case):
    """the same in the frames-in-flight path: the frame whose lane overflowed is flagged in its own
    header, the others complete normally and equal the serial result"""
    from blackbox_amd.pipeline import FramePipeline, HostPool
    geom = R.geometry(case['raw'].shape, YS, XS)
    coeffs = O.xtalk_coeffs(case['xtalk'])
    pool = HostPool(2)
    pipe = FramePipeline(ctx, TEL, geom, mflat=dev(ctx, case['flat']), bpm=dev(ctx, case['bpm']), xtalk_coeffs=coeffs,
                         exptime=60.0, pool=pool, depth=3, lanes=2, do_finish=True, detect_sats=True, keep_outputs=True)
    raw = dev(ctx, case['raw'])
    done = {}
    try:
        # lane 1's context gets the tiny list capacity: frames 1, 3 fail their cosmics step
        _lib.check(_lib.lib.bbx_set_option(pipe.lane_ctx[1].h, 2, 8), 'bbx_set_option')
        pipe.run([(raw, {}) for _ in range(4)], on_done=lambda i, f: done.__setitem__(i, (f.header, f.hm, f.data.clone(), f.mask.clone(), list(f.failed))))
    finally:
        pipe.close()
        pool.close()
    want = run(ctx, case)
    for i in (0, 2):
        h, hm, d, m, failed = done[i]
        assert failed == [] and R.hval(h, 'COSMIC-P') is True
        assert torch.equal(d, want[0]) and torch.equal(m, want[1])
        assert R.hval(h, 'NCOSMICS') == R.hval(want[2], 'NCOSMICS') and R.hval(h, 'NSATS') == R.hval(want[2], 'NSATS')
        assert hm['M-CRPNUM'] == want[3]['M-CRPNUM'] and hm['M-SPNUM'] == want[3]['M-SPNUM']
    for i in (1, 3):
        h, hm, d, m, failed = done[i]
        assert 'cosmics' in failed and R.hval(h, 'COSMIC-P') is False and R.hval(h, 'NCOSMICS') == 'None'
        assert R.hval(h, 'XTALK-P') is True and R.hval(h, 'SAT-P') is True


def test_flat_frame_mode(ctx, case):
    """imgtype 'flat' (blackbox.py:1749-1784): mask = the bad-pixel mask as it is, no saturation
    marking, no flat division / cosmics / edge fill"""
    bpm = dev(ctx, case['bpm'])
    d, m, h, hm = R.reduce_object(ctx, dev(ctx, case['raw']), {}, TEL, mflat=dev(ctx, case['flat']), bpm=bpm, exptime=5.0,
                                  ysize_chan=YS, xsize_chan=XS, imgtype='flat')
    assert torch.equal(m, bpm) and 'SATURATE' not in h and 'COSMIC-P' not in h and 'MFLAT-P' not in h
    o = case['raw'].astype(np.float32)
    O.gain_corr(o, settings.gain[TEL], YS, XS)
    o, oh, _ = O.os_corr(o, YS, XS, tel=TEL, accum='bn32')
    assert np.array_equal(d.cpu().numpy(), o)                    # overscan-corrected only: not flat-fielded, edges untouched


def moffat(S, fwhm):
    a = fwhm / (2 * np.sqrt(2 ** (1 / 2.5) - 1))
    y, x = np.mgrid[0:S, 0:S] - S // 2
    p = (1 + (y * y + x * x) / (a * a)) ** -2.5
    return (p / p.sum()).astype(np.float32)


def test_cli_subtraction_products_and_image_list(tmp_path, ctx, case):
    cli = load_cli()
    hdr = {'EXPTIME': 60.0, 'IMAGETYP': 'object', 'FILTER': 'q'}
    raws = []
    for k in range(3):
        p = str(tmp_path / ('ML1_raw%d.fits' % k))                   # the same field three times (one reference image)
        fitsio.write_image(p, case['raw'], dict(hdr, **{'DATE-OBS': '2024-01-02T03:04:0%d' % k}))
        raws.append(p)
    fitsio.write_image(str(tmp_path / 'flat.fits'), case['flat'])
    fitsio.write_image(str(tmp_path / 'bpm.fits'), case['bpm'])
    synth.write_xtalk(str(tmp_path / 'xtalk.dat'), case['xtalk'])
    d0, m0, h0, _ = run(ctx, case)
    rs = np.random.RandomState(3)
    ref = (d0.cpu().numpy() - 100.0 + rs.normal(0, 4, d0.shape)).astype(np.float32)
    fitsio.write_image(str(tmp_path / 'ref.fits'), ref)
    fitsio.write_image(str(tmp_path / 'psf.fits'), moffat(15, 3.5))
    common = ['--telescope', TEL, '--mflat', str(tmp_path / 'flat.fits'), '--bpm', str(tmp_path / 'bpm.fits'),
              '--crosstalk', str(tmp_path / 'xtalk.dat'), '--ysize_chan', str(YS), '--xsize_chan', str(XS),
              '--cat_extract', 'True', '--trans_extract', 'True', '--ref', str(tmp_path / 'ref.fits'),
              '--psf_new', str(tmp_path / 'psf.fits'), '--psf_ref', str(tmp_path / 'psf.fits'),
              '--subimage_size', '120', '--subimage_border', '10', '--bkg_boxsize', '30']
    out = cli.main(common + ['--image', raws[0], '--red_dir', str(tmp_path / 'a')])
    base = str(tmp_path / 'a' / 'ML1_20240102_030400_red')
    assert out == [base + '.fits']
    h = fitsio.read_image(base + '.fits', get_header=True)[1]
    assert R.hval(h, 'QC-FLAG') != 'red', {k: R.hval(h, k) for k in h if k.startswith('QC')}
    for ext in ('.fits', '_hdr.fits', '.log', '_bkg_mini.fits', '_bkg_std_mini.fits', '_cat.fits', '_cat_hdr.fits', '_D.fits',
                '_Scorr.fits', '_Fpsf.fits', '_trans.fits', '_trans_hdr.fits'):
        assert os.path.isfile(base + ext), ext
    assert os.path.isfile(base.replace('_red', '_mask') + '.fits')
    assert R.hval(h, 'Z-P') is True and R.hval(h, 'S-BKG') > 0 and R.hval(h, 'BKG-SIZE') == 30
    # the images on disk are what the device function returns
    psf = dev(ctx, moffat(15, 3.5))
    res = G.optimal_subtraction(ctx, d0, dev(ctx, ref), m0, torch.zeros_like(m0), psf, psf, subimage_size=120,
                                subimage_border=10, bkg_boxsize=30, cat_extract=True)
    ctx.sync()
    for ext, key in (('_D', 'D'), ('_Scorr', 'Scorr'), ('_Fpsf', 'Fpsf')):
        assert np.array_equal(fitsio.read_image(base + ext + '.fits'), res[key].cpu().numpy(), equal_nan=True), ext
    assert np.array_equal(fitsio.read_image(base + '_bkg_mini.fits'), res['bkg_mini_new'])
    cat, hc = fitsio.read_table(base + '_cat.fits')
    assert len(cat['X_POS']) == len(res['catalog']['X_POS']) > 10 and np.array_equal(cat['E_FLUX_OPT'], res['catalog']['E_FLUX_OPT'])
    tr, ht = fitsio.read_table(base + '_trans.fits')
    assert len(tr['X_PEAK']) == len(res['transients']) == R.hval(ht, 'T-NTRANS')
    hh = fitsio.read_hdus(base + '_hdr.fits')[0][0]
    assert R.hval(hh, 'RDNOISE') == R.hval(h, 'RDNOISE') and R.hval(hh, 'Z-P') is True
    # `_trans_limmag`: T-NSIGMA x Fpsferr; a flux without a zeropoint, magnitudes with one
    lim, hl = fitsio.read_image(base + '_trans_limmag.fits', get_header=True)
    assert R.hval(hl, 'LIMUNIT') == 'e-' and R.hval(hl, 'LIMNSIG') == 6.0
    assert np.array_equal(lim, (res['Fpsferr'] * 6.0).cpu().numpy(), equal_nan=True)
    cli.main(common + ['--image', raws[0], '--red_dir', str(tmp_path / 'z'), '--zeropoint', '22.5'])
    mag, hl = fitsio.read_image(str(tmp_path / 'z' / 'ML1_20240102_030400_red_trans_limmag.fits'), get_header=True)
    ok = lim > 0
    assert R.hval(hl, 'LIMUNIT') == 'mag'
    np.testing.assert_allclose(mag[ok], 22.5 - 2.5 * np.log10(lim[ok] / 60.0), rtol=0, atol=2e-5)

    # --image_list: the same frames through the frames-in-flight pipeline
    lst = str(tmp_path / 'list.txt')
    with open(lst, 'w') as f:
        f.write('\n'.join(raws) + '\n')
    outs = cli.main(common + ['--image_list', lst, '--red_dir', str(tmp_path / 'b')])
    assert len(outs) == 3 and all(o and os.path.isfile(o) for o in outs)
    b0 = str(tmp_path / 'b' / 'ML1_20240102_030400_red')
    for ext in ('.fits', '_D.fits', '_Scorr.fits', '_Fpsf.fits', '_bkg_std_mini.fits'):
        assert np.array_equal(fitsio.read_image(b0 + ext), fitsio.read_image(base + ext), equal_nan=True), ext
    assert np.array_equal(fitsio.read_image(b0.replace('_red', '_mask') + '.fits'),
                          fitsio.read_image(base.replace('_red', '_mask') + '.fits'))
    hb = fitsio.read_image(b0 + '.fits', get_header=True)[1]
    for k in ('RDNOISE', 'NCOSMICS', 'NSATS', 'NOBJ-SAT', 'S-BKG', 'Z-P', 'COSMIC-P'):
        assert R.hval(hb, k) == R.hval(h, k), k
    for k in (1, 2):
        assert os.path.isfile(str(tmp_path / 'b' / ('ML1_20240102_03040%d_red_trans.fits' % k)))


FARM_SCRIPT = """
import json, sys
sys.path.insert(0, {root!r})
import blackbox
if __name__ == '__main__':
    argv, files = json.loads(sys.argv[1]), json.loads(sys.argv[2])
    blackbox.configure(argv)
    import torch
    assert not torch.cuda.is_initialized()
    try:
        out = blackbox.pool_func(blackbox.try_blackbox_reduce, files, nproc=2)
        assert not torch.cuda.is_initialized()                    # the parent never touched the GPU
        print('OUT ' + json.dumps(out))
    except blackbox.WrapException as e:
        print('WRAPPED ' + json.dumps(e.formatted))
"""


def run_plain(ctx, case):
    return R.reduce_object(ctx, dev(ctx, case['raw']), {}, TEL, mflat=dev(ctx, case['flat']), bpm=dev(ctx, case['bpm']),
                           exptime=60.0, ysize_chan=YS, xsize_chan=XS)


def test_farm_spawn_pool_and_wrapexception(tmp_path, case):
    """the operator-level entry the reference's farm imports (blackbox.py:363-379, 933-999): module-level
    try_blackbox_reduce mapped over 3 files by a 2-worker multiprocessing SPAWN pool -- every child creates its own
    GPU context, the parent (a fresh interpreter here) makes no GPU call -- and a file that makes blackbox_reduce
    raise comes back as WrapException with the worker's formatted traceback"""
    import json
    import subprocess
    hdr = {'EXPTIME': 60.0, 'IMAGETYP': 'object', 'FILTER': 'q'}
    raws = []
    for k in range(3):
        p = str(tmp_path / ('ML1_raw%d.fits' % k))
        fitsio.write_image(p, case['raw'], dict(hdr, **{'DATE-OBS': '2024-01-02T03:04:0%d' % k}))
        raws.append(p)
    fitsio.write_image(str(tmp_path / 'flat.fits'), case['flat'])
    fitsio.write_image(str(tmp_path / 'bpm.fits'), case['bpm'])
    argv = ['--telescope', TEL, '--mflat', str(tmp_path / 'flat.fits'), '--bpm', str(tmp_path / 'bpm.fits'),
            '--ysize_chan', str(YS), '--xsize_chan', str(XS), '--red_dir', str(tmp_path / 'p')]
    script = str(tmp_path / 'farm.py')
    with open(script, 'w') as f:
        f.write(FARM_SCRIPT.format(root=os.path.abspath(ROOT)))
    r = subprocess.run([sys.executable, script, json.dumps(argv), json.dumps(raws)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('OUT ')]
    assert line, (r.stdout[-2000:], r.stderr[-2000:])
    outs = json.loads(line[0][4:])
    assert [os.path.basename(o) for o in outs] == ['ML1_20240102_03040%d_red.fits' % k for k in range(3)]
    # the products of the pool equal the in-process reduction
    c = R.Context(0)
    try:
        want = run_plain(c, case)
        for o in outs:
            assert np.array_equal(fitsio.read_image(o), want[0].cpu().numpy())
            assert np.array_equal(fitsio.read_image(o.replace('_red', '_mask')), want[1].cpu().numpy())
    finally:
        c.close()
    # a file blackbox_reduce cannot read: WrapException with the traceback text reaches the parent
    bad = str(tmp_path / 'ML1_missing.fits')
    r = subprocess.run([sys.executable, script, json.dumps(argv), json.dumps([raws[0], bad])], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('WRAPPED ')]
    assert line, (r.stdout[-2000:], r.stderr[-2000:])
    tb = json.loads(line[0][8:])
    assert 'Traceback' in tb and 'ML1_missing.fits' in tb and 'blackbox_reduce' in tb


def test_cli_image_list_through_output_stage(tmp_path, ctx, case):
    """--image_list --fpack True: the image products are compressed on the lane that made them and written by the output
    stage's threads (blackbox_amd/outstage.py), headers completed by the serial path's own code once the frame's scalars
    are in.  Same files as the one-by-one --image --fpack True run: identical compressed tables + heaps, identical
    header keywords (but for the time stamps), the small products next to them."""
    cli = load_cli()
    hdr = {'EXPTIME': 60.0, 'IMAGETYP': 'object', 'FILTER': 'q'}
    raws = []
    for k in range(3):
        p = str(tmp_path / ('ML1_raw%d.fits' % k))
        fitsio.write_image(p, case['raw'], dict(hdr, **{'DATE-OBS': '2024-01-02T03:04:0%d' % k}))
        raws.append(p)
    fitsio.write_image(str(tmp_path / 'flat.fits'), case['flat'])
    fitsio.write_image(str(tmp_path / 'bpm.fits'), case['bpm'])
    synth.write_xtalk(str(tmp_path / 'xtalk.dat'), case['xtalk'])
    d0, m0, h0, _ = run(ctx, case)
    rs = np.random.RandomState(3)
    fitsio.write_image(str(tmp_path / 'ref.fits'), (d0.cpu().numpy() - 100.0 + rs.normal(0, 4, d0.shape)).astype(np.float32))
    fitsio.write_image(str(tmp_path / 'psf.fits'), moffat(15, 3.5))
    common = ['--telescope', TEL, '--mflat', str(tmp_path / 'flat.fits'), '--bpm', str(tmp_path / 'bpm.fits'),
              '--crosstalk', str(tmp_path / 'xtalk.dat'), '--ysize_chan', str(YS), '--xsize_chan', str(XS),
              '--cat_extract', 'True', '--trans_extract', 'True', '--ref', str(tmp_path / 'ref.fits'),
              '--psf_new', str(tmp_path / 'psf.fits'), '--psf_ref', str(tmp_path / 'psf.fits'),
              '--subimage_size', '120', '--subimage_border', '10', '--bkg_boxsize', '30', '--fpack', 'True']
    one = cli.main(common + ['--image', raws[0], '--red_dir', str(tmp_path / 'one')])
    lst = str(tmp_path / 'list.txt')
    with open(lst, 'w') as f:
        f.write('\n'.join(raws) + '\n')
    outs = cli.main(common + ['--image_list', lst, '--red_dir', str(tmp_path / 'lst')])
    assert len(outs) == 3 and all(o and o.endswith('_red.fits.fz') and os.path.isfile(o) for o in outs), outs
    assert os.path.basename(outs[0]) == os.path.basename(one[0])
    b1, b2 = one[0].replace('.fits.fz', ''), outs[0].replace('.fits.fz', '')
    for ext in ('.fits.fz', '_D.fits.fz', '_Scorr.fits.fz', '_Fpsf.fits.fz', '_trans_limmag.fits.fz'):
        (_, _), (ha, ta) = fitsio.read_hdus(b1 + ext)
        (_, _), (hb, tb) = fitsio.read_hdus(b2 + ext)
        assert np.array_equal(ta, tb) and np.array_equal(ha['__heap__'], hb['__heap__']), ext
        ka = {k: R.hval(ha, k) for k in ha if k not in ('__heap__', 'BB-START')}
        kb = {k: R.hval(hb, k) for k in hb if k not in ('__heap__', 'BB-START')}
        assert ka == kb, (ext, {k: (ka.get(k), kb.get(k)) for k in set(ka) | set(kb) if ka.get(k) != kb.get(k)})
        assert R.hval(hb, 'ZCMPTYPE') == 'RICE_1'
    ma, mb = fitsio.read_hdus(b1.replace('_red', '_mask') + '.fits.fz')[1], fitsio.read_hdus(b2.replace('_red', '_mask') + '.fits.fz')[1]
    assert np.array_equal(ma[1], mb[1]) and np.array_equal(ma[0]['__heap__'], mb[0]['__heap__'])
    for ext in ('_hdr.fits', '_cat.fits', '_trans.fits', '_trans_hdr.fits', '_bkg_mini.fits', '.log'):
        assert os.path.isfile(b2 + ext), ext
    for k in (1, 2):
        assert os.path.isfile(str(tmp_path / 'lst' / ('ML1_20240102_03040%d_red_Scorr.fits.fz' % k)))


def test_cli_image_list_one_bad_file_fails_one_file(tmp_path, ctx, case):
    """blackbox.py:948-999: an exception while a file is reduced is logged, that file yields None, the list goes on.  A
    six-file --image_list with a raw frame cut short (still being written) and a frame of another shape in it: four sets of
    products, two None, through the pipeline (input stage, lanes, output stage) -- which is not restarted and reduces
    nothing one by one."""
    cli = load_cli()
    hdr = {'EXPTIME': 60.0, 'IMAGETYP': 'object', 'FILTER': 'q'}
    raws = []
    for k in range(6):
        p = str(tmp_path / ('ML1_raw%d.fits' % k))
        img = case['raw'] if k != 4 else case['raw'][:-2]                  # file 4: another shape
        fitsio.write_image(p, img, dict(hdr, **{'DATE-OBS': '2024-01-02T03:04:0%d' % k}))
        raws.append(p)
    n = os.path.getsize(raws[1])
    with open(raws[1], 'r+b') as f:                                        # file 1: the last third of the pixels missing
        f.truncate(n - n // 3)
    fitsio.write_image(str(tmp_path / 'flat.fits'), case['flat'])
    fitsio.write_image(str(tmp_path / 'bpm.fits'), case['bpm'])
    synth.write_xtalk(str(tmp_path / 'xtalk.dat'), case['xtalk'])
    lst = str(tmp_path / 'list.txt')
    with open(lst, 'w') as f:
        f.write('\n'.join(raws) + '\n')
    calls = []
    orig = cli.Reducer.reduce_logged

    def spy(self, fn):
        calls.append(fn)
        return orig(self, fn)
    cli.Reducer.reduce_logged = spy
    try:
        outs = cli.main(['--telescope', TEL, '--mflat', str(tmp_path / 'flat.fits'), '--bpm', str(tmp_path / 'bpm.fits'),
                         '--crosstalk', str(tmp_path / 'xtalk.dat'), '--ysize_chan', str(YS), '--xsize_chan', str(XS), '--fpack', 'True',
                         '--image_list', lst, '--red_dir', str(tmp_path / 'red')])
    finally:
        cli.Reducer.reduce_logged = orig
    assert len(outs) == 6 and outs[1] is None and outs[4] is None, outs
    good = [outs[k] for k in (0, 2, 3, 5)]
    assert all(o and o.endswith('_red.fits.fz') and os.path.isfile(o) and os.path.isfile(o.replace('_red', '_mask')) for o in good), outs
    assert not calls                                                       # nothing went through the one-by-one path
    from blackbox_amd import fpack as P
    _, m0, _, _ = run(ctx, case)
    for o in good:
        got = P.funpack_image(ctx, o.replace('_red', '_mask'))
        got = got[0] if isinstance(got, tuple) else got
        assert np.array_equal(np.asarray(got.cpu() if hasattr(got, 'cpu') else got), m0.cpu().numpy()), o


def test_cli_image_list_in_two_processes(tmp_path, ctx, case):
    """--image_list --list_procs 2: two pipelined child processes share the GPU, the files of the list alternate between
    them.  Results come back in the order of the list, a file that cannot be read is None in its place, and the products are
    the bytes the one-process list makes."""
    cli = load_cli()
    hdr = {'EXPTIME': 60.0, 'IMAGETYP': 'object', 'FILTER': 'q'}
    raws = []
    for k in range(5):
        p = str(tmp_path / ('ML1_raw%d.fits' % k))
        fitsio.write_image(p, case['raw'], dict(hdr, **{'DATE-OBS': '2024-01-02T03:04:0%d' % k}))
        raws.append(p)
    n = os.path.getsize(raws[3])
    with open(raws[3], 'r+b') as f:                                        # file 3 (second child's): cut short
        f.truncate(n - n // 3)
    fitsio.write_image(str(tmp_path / 'flat.fits'), case['flat'])
    fitsio.write_image(str(tmp_path / 'bpm.fits'), case['bpm'])
    synth.write_xtalk(str(tmp_path / 'xtalk.dat'), case['xtalk'])
    lst = str(tmp_path / 'list.txt')
    with open(lst, 'w') as f:
        f.write('\n'.join(raws) + '\n')
    common = ['--telescope', TEL, '--mflat', str(tmp_path / 'flat.fits'), '--bpm', str(tmp_path / 'bpm.fits'),
              '--crosstalk', str(tmp_path / 'xtalk.dat'), '--ysize_chan', str(YS), '--xsize_chan', str(XS), '--fpack', 'True',
              '--image_list', lst]
    one = cli.main(common + ['--red_dir', str(tmp_path / 'p1'), '--list_procs', '1'])
    two = cli.main(common + ['--red_dir', str(tmp_path / 'p2'), '--list_procs', '2'])
    assert len(two) == 5 and two[3] is None and one[3] is None, (one, two)
    for k in (0, 1, 2, 4):
        assert two[k] and os.path.basename(two[k]) == os.path.basename(one[k]) and os.path.isfile(two[k]), (k, two)
        for a, b in ((one[k], two[k]), (one[k].replace('_red', '_mask'), two[k].replace('_red', '_mask'))):
            (_, _), (ha, ta) = fitsio.read_hdus(a)
            (_, _), (hb, tb) = fitsio.read_hdus(b)
            assert np.array_equal(ta, tb) and np.array_equal(ha['__heap__'], hb['__heap__']), (k, a)
        assert os.path.isfile(two[k].replace('.fits.fz', '_hdr.fits'))
