"""GPU: the input stage (blackbox_amd/instage.py; reference: read_hdulist, blackbox.py:1451) -- raw frames read from
`.fits.fz` / `.fits` files by reader threads, decoded on the device, handed over with ready events: the frames equal what
was written, in file order, and the pipeline fed from files gives the very results of the pipeline fed from HBM."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
if not torch.cuda.is_available():
    pytest.skip('no GPU', allow_module_level=True)

import bbx_oracle as O                                  # noqa: E402
from blackbox_amd import fitsio, instage, synth         # noqa: E402
from blackbox_amd import fpack as P                     # noqa: E402
from blackbox_amd import reduce as R                    # noqa: E402
from blackbox_amd.pipeline import FramePipeline, HostPool   # noqa: E402


def test_input_stage_and_pipeline_from_files(tmp_path):
    ctx = R.Context(0)
    dev = ctx.device
    tel, ys, xs, os_y, os_x, nframes = 'ML1', 96, 330, 20, 45, 7
    cases = [synth.make_case(ys, xs, 300 + k, tel=tel, os_y=os_y, os_x=os_x, n_stars=60, n_sat=4, n_cr=60) for k in range(nframes)]
    raws = [torch.from_numpy(c['raw']).to(dev) for c in cases]
    assert raws[0].dtype == torch.uint16
    files = []
    for k, rw in enumerate(raws):
        hdr = {'EXPTIME': 60.0, 'OBJECT': 'frame%d' % k, 'IMAGETYP': 'object'}
        if k % 3 == 2:                                        # every third file uncompressed
            p = str(tmp_path / ('raw%d.fits' % k))
            fitsio.write_image(p, cases[k]['raw'], hdr)
        else:
            p = P.fpack_image(ctx, str(tmp_path / ('raw%d.fits' % k)), rw, hdr)
        files.append(p)
    ctx.sync()
    # (i) the stage alone: order, pixels, headers; the pool of raw buffers is smaller than the number of files
    st = instage.InputStage(ctx, files, tuple(raws[0].shape), nreaders=3, nbuf=4, ahead=2)
    n = 0
    for k, (raw, header, ev) in enumerate(st):
        ev.synchronize()
        assert torch.equal(raw, raws[k]), k
        assert fitsio._hv(header, 'OBJECT') == 'frame%d' % k and fitsio._hv(header, 'EXPTIME') == 60.0
        assert 'BZERO' not in header and 'ZIMAGE' not in header and 'NAXIS1' not in header
        st.release(raw)
        n += 1
    assert n == nframes and st.bytes_read == sum(os.path.getsize(f) for f in files)
    st.close()
    # a missing file surfaces at its place in the order
    st = instage.InputStage(ctx, files[:2] + [str(tmp_path / 'nope.fits.fz')], tuple(raws[0].shape), nreaders=2, nbuf=3, ahead=2)
    it = iter(st)
    for _ in range(2):
        raw, _, ev = next(it)
        ev.synchronize()
        st.release(raw)
    with pytest.raises(OSError):
        next(it)
    st.close()
    # (ii) the pipeline fed from the files == the pipeline fed from HBM
    flat = torch.from_numpy(cases[0]['flat']).to(dev)
    bpm = torch.from_numpy(cases[0]['bpm']).to(dev)
    coeffs = O.xtalk_coeffs(cases[0]['xtalk'])
    geom = R.geometry(raws[0].shape, ys, xs)
    pool = HostPool(3)
    try:
        out = {}
        for mode in ('hbm', 'files'):
            pipe = FramePipeline(ctx, tel, geom, mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0, pool=pool, depth=3, lanes=2,
                                 do_finish=True, keep_outputs=True)
            res = {}
            st = instage.InputStage(ctx, files, tuple(raws[0].shape), nreaders=2, nbuf=3 + 2, ahead=2) if mode == 'files' else None

            def on_done(idx, f):
                res[idx] = (f.data.cpu().numpy(), f.mask.cpu().numpy(), dict(f.header))
                if st is not None:
                    st.release(f.raw)
            nd = pipe.run(st if st is not None else [(rw, {}) for rw in raws], on_done=on_done)
            assert nd == nframes
            pipe.close()
            if st is not None:
                st.close()
            out[mode] = res
        for k in range(nframes):
            a, b = out['hbm'][k], out['files'][k]
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), k
            for key in ('BIASMEAN', 'RDNOISE', 'NCOSMICS', 'NOBJ-SAT'):
                assert R.hval(a[2], key) == R.hval(b[2], key), (k, key)
            assert R.hval(b[2], 'OBJECT') == 'frame%d' % k                     # the file's header went along
    finally:
        pool.close()
    ctx.close()
