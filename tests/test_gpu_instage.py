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
    # one bad file fails one file (blackbox.py:948-999): a missing file, an fpacked file cut short (still being written), a frame
    # of another shape -- each surfaces as instage.InputError at its place in the order, the frames behind it follow, the ONE
    # reader thread lives on and every raw buffer is back in the pool
    cut = str(tmp_path / 'cut.fits.fz')
    whole = open(files[0], 'rb').read()
    with open(cut, 'wb') as f:
        f.write(whole[:len(whole) - 4000])
    other = str(tmp_path / 'other.fits')
    fitsio.write_image(other, cases[0]['raw'][:-2], {'EXPTIME': 60.0})
    mixed = [files[0], str(tmp_path / 'nope.fits.fz'), files[1], cut, files[2], other, files[3]]
    st = instage.InputStage(ctx, mixed, tuple(raws[0].shape), nreaders=1, nbuf=3, ahead=2)
    it = iter(st)
    got, errs = [], []
    for k in range(len(mixed)):
        try:
            raw, _, ev = next(it)
        except instage.InputError as e:
            errs.append((k, e.idx, type(e.cause)))
            continue
        ev.synchronize()
        got.append((k, raw.clone()))
        st.release(raw)
    with pytest.raises(StopIteration):
        next(it)
    assert [k for k, *_ in errs] == [1, 3, 5] and [i for _, i, _ in errs] == [1, 3, 5]
    assert issubclass(errs[0][2], OSError) and errs[1][2] is EOFError and errs[2][2] is ValueError
    assert [k for k, _ in got] == [0, 2, 4, 6]
    for (k, rw), want in zip(got, (raws[0], raws[1], raws[2], raws[3])):
        assert torch.equal(rw, want), k
    assert st.pool.qsize() == 3          # (the ONE reader delivered the frames behind every bad file: it lived on; it ends with the list)
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
        # the pipeline on the list with the bad files in it: their indices are skipped (the hook hears of them), the others come
        # out as from the clean list, nothing is restarted
        pipe = FramePipeline(ctx, tel, geom, mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0, pool=pool, depth=3, lanes=2,
                             do_finish=True, keep_outputs=True)
        st = instage.InputStage(ctx, mixed, tuple(raws[0].shape), nreaders=2, nbuf=5, ahead=2)
        res, bad = {}, {}

        def on_done2(idx, f):
            res[idx] = (f.data.cpu().numpy(), f.mask.cpu().numpy())
            st.release(f.raw)
        nd = pipe.run(st, on_done=on_done2, on_input_error=lambda idx, e: bad.__setitem__(idx, e))
        assert nd == 4 and sorted(res) == [0, 2, 4, 6] and sorted(bad) == [1, 3, 5]
        for idx, k in ((0, 0), (2, 1), (4, 2), (6, 3)):
            assert np.array_equal(res[idx][0], out['hbm'][k][0]) and np.array_equal(res[idx][1], out['hbm'][k][1]), idx
        # without the hook the first bad file ends the run (the old contract), and the pipeline is usable afterwards
        st2 = instage.InputStage(ctx, mixed[:3], tuple(raws[0].shape), nreaders=2, nbuf=5, ahead=2)
        with pytest.raises(instage.InputError):
            pipe.run(st2, on_done=lambda idx, f: st2.release(f.raw))
        st2.close()
        assert pipe.run([(raws[0], {})]) == 1
        pipe.close()
        st.close()
    finally:
        pool.close()
    ctx.close()


def test_master_frames_go_up_in_file_order_and_are_swapped_on_the_device(tmp_path):
    """read_hdulist of a master flat / bias / reference image (blackbox.py:1677, 1823): reduce.image_to_device uploads the
    file's big-endian bytes and bbx_be32 puts them into host order -- the very bits fitsio.read_image makes on the host,
    for sizes that are and are not a multiple of the kernel's 16-byte groups; uint8 masks go up as they are; a file with
    BZERO takes the host path."""
    ctx = R.Context(0)
    rs = np.random.RandomState(4)
    for shape in ((33, 47), (64, 128), (1, 5), (257, 1023)):
        a = (rs.standard_normal(shape) * 10.0 ** rs.uniform(-20, 20, shape)).astype(np.float32)
        a.flat[0] = np.float32(-0.0); a.flat[-1] = np.float32(np.inf)
        p = str(tmp_path / ('f_%dx%d.fits' % shape))
        fitsio.write_image(p, a)
        assert fitsio.read_image_file_order(p)[0].dtype == np.dtype('>f4')
        got = R.image_to_device(ctx, p, np.float32)
        ctx.sync()
        assert got.dtype == torch.float32 and tuple(got.shape) == shape
        assert np.array_equal(got.cpu().numpy().view(np.uint32), fitsio.read_image(p, dtype=np.float32).view(np.uint32)), shape
        assert np.array_equal(got.cpu().numpy().view(np.uint32), a.view(np.uint32))
    m = (rs.rand(40, 50) < 0.1).astype(np.uint8) * 32
    p = str(tmp_path / 'm.fits')
    fitsio.write_image(p, m)
    got = R.image_to_device(ctx, p, np.uint8)
    assert got.dtype == torch.uint8 and np.array_equal(got.cpu().numpy(), m)
    # an int16 file read as float32, and unsigned 16-bit (BZERO 32768): converted on the host as before
    p = str(tmp_path / 'i.fits')
    fitsio.write_image(p, rs.randint(-100, 100, (12, 13)).astype(np.int16))
    got = R.image_to_device(ctx, p, np.float32)
    assert np.array_equal(got.cpu().numpy(), fitsio.read_image(p, dtype=np.float32))
    p = str(tmp_path / 'u.fits')
    u = rs.randint(0, 65535, (12, 13)).astype(np.uint16)
    fitsio.write_image(p, u)
    assert fitsio.read_image_file_order(p) is None
    assert np.array_equal(R.image_to_device(ctx, p, np.float32).cpu().numpy(), u.astype(np.float32))
    # unaligned views of a device buffer (the one-word tail and the scalar path of the kernel)
    from blackbox_amd._lib import lib, check
    import ctypes as C
    w = torch.from_numpy(rs.randint(0, 2 ** 31 - 1, 1001).astype(np.int32)).to(ctx.device)
    for off, n in ((0, 1001), (1, 1000), (3, 5), (4, 997)):
        src = w[off:off + n]
        dst = torch.empty(n + 1, dtype=torch.int32, device=ctx.device)[1:] if off == 1 else torch.empty(n, dtype=torch.int32, device=ctx.device)
        check(lib.bbx_be32(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, ctx.stream()), 'bbx_be32')
        ctx.sync()
        assert np.array_equal(dst.cpu().numpy(), src.cpu().numpy().byteswap()), (off, n)
    ctx.close()
