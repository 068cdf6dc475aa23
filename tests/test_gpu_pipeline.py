"""GPU: the pipelined throughput path (FramePipeline: device statistics -> host fit workers
over the shared pinned arena -> device stage) must give exactly what the serial
reduce_object gives for the same frames, for the one-phase (ML1) and the two-phase
(BlackGEM, saturated-column step) overscan solve."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
if not torch.cuda.is_available():
    pytest.skip('no GPU', allow_module_level=True)

import bbx_oracle as O                                  # noqa: E402
from blackbox_amd import reduce as R                    # noqa: E402
from blackbox_amd import synth                          # noqa: E402
from blackbox_amd.pipeline import FramePipeline, HostPool   # noqa: E402


@pytest.fixture(scope='module')
def pool():
    p = HostPool(4)
    yield p
    p.close()


@pytest.mark.parametrize('tel,ys,xs,os_y,os_x,lanes,nframes', [('ML1', 96, 330, 20, 45, 2, 4), ('BG3', 2640, 330, 20, 45, 2, 4),
                                                               ('ML1', 96, 330, 20, 45, 6, 12)])
def test_pipeline_equals_serial(pool, tel, ys, xs, os_y, os_x, lanes, nframes):
    """(the six-lane case: twelve frames of which every other one needs LA-Cosmic's background level)"""
    ctx = R.Context(0)
    dev = ctx.device
    cases = [synth.make_case(ys, xs, 100 + k, tel=tel, os_y=os_y, os_x=os_x, n_stars=60, n_sat=4, n_cr=60)
             for k in range(nframes)]
    # frames 1 and 3 hold a hot pixel inside a fully masked 5x5 block: the cleaned value is LA-Cosmic's
    # background level -- selected over the frame on demand in the serial runs and in the pipeline's
    # first such frame, prepared in advance (BBX_OPT_LAC_LEVEL_FEED) in the pipeline's later ones
    bpm_np = cases[0]['bpm'].copy()
    hot = [(ys // 2 + 7, xs + 13), (ys + 5, 3 * xs + 40)]
    for (j, i) in hot:
        bpm_np[j - 2:j + 3, i - 2:i + 3] |= 1
        bpm_np[j, i] = 0
    dy, dx = ys + os_y, xs + os_x
    for k in range(1, nframes, 2):
        for (j, i) in hot:
            iy, ix = j // ys, i // xs
            rj = iy * dy + (j - iy * ys) + (0 if iy == 0 else os_y)
            cases[k]['raw'][rj, ix * dx + (i - ix * xs)] = 15000       # well below saturation
    flat = torch.from_numpy(cases[0]['flat']).to(dev)
    bpm = torch.from_numpy(bpm_np).to(dev)
    coeffs = O.xtalk_coeffs(cases[0]['xtalk'])
    raws = [torch.from_numpy(c['raw']).to(dev) for c in cases]
    geom = R.geometry(raws[0].shape, ys, xs)

    serial = []
    for raw in raws:
        d, m, h, hm = R.reduce_object(ctx, raw, {}, tel, mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0,
                                      ysize_chan=ys, xsize_chan=xs, detect_sats=False)
        serial.append((d.cpu().numpy(), m.cpu().numpy(), h))

    pipe = FramePipeline(ctx, tel, geom, mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0, pool=pool,
                         depth=3 if lanes == 2 else 9, do_finish=True, keep_outputs=True, lanes=lanes)
    got = {}

    def done(idx, f):
        got[idx] = (f.data.cpu().numpy(), f.mask.cpu().numpy(), f.header)
    n = pipe.run([(r, {}) for r in raws], on_done=done)
    fed = pipe.level_feed_left
    pipe.close()
    assert fed > 0                                                # a frame needed the level: the feed is on now
    assert n == len(raws) and sorted(got) == list(range(len(raws)))
    for k in range(len(raws)):
        d0, m0, h0 = serial[k]
        d1, m1, h1 = got[k]
        assert np.array_equal(m0, m1), 'frame %d: mask' % k
        assert np.array_equal(d0, d1), 'frame %d: pixels' % k
        keys = ['BIASMEAN', 'RDNOISE', 'NOBJ-SAT', 'NCOSMICS', 'N-INFNAN'] + \
               ['%s%d' % (p, c + 1) for p in ('BIASM', 'RDN', 'VFITOK') for c in range(16)] + \
               ['BIAS%dA%d' % (c + 1, j) for c in range(16) for j in range(4)]
        for key in keys:
            assert R.hval(h0, key) == R.hval(h1, key), (k, key)
    ctx.close()


def test_pipeline_refuses_bad_input_and_reports_lane_errors(pool):
    """a frame of the wrong shape is refused before anything is launched; an exception raised while a
    lane thread issues the device stage comes out of run()"""
    ctx = R.Context(0)
    ys, xs = 96, 330
    case = synth.make_case(ys, xs, 7, tel='ML1', os_y=20, os_x=45, n_stars=10, n_sat=1, n_cr=5)
    raw = torch.from_numpy(case['raw']).to(ctx.device)
    geom = R.geometry(raw.shape, ys, xs)
    pipe = FramePipeline(ctx, 'ML1', geom, pool=pool, depth=2, lanes=2)
    with pytest.raises(ValueError):
        pipe.run([(raw[:-1].contiguous(), {})])
    assert pipe.run([(raw, {})]) == 1                              # still usable afterwards
    pipe.close()
    bad_flat = torch.ones((2 * ys, 8 * xs - 1), dtype=torch.float32, device=ctx.device)
    pipe = FramePipeline(ctx, 'ML1', geom, mflat=bad_flat, pool=pool, depth=2, lanes=2)
    with pytest.raises(ValueError):
        pipe.run([(raw, {}), (raw, {})])
    assert sorted(pipe.free_slots) == [0, 1]                       # nothing leaked by the aborted run
    pipe.close()
    ctx.close()


def test_two_threads_wait_on_one_context():
    """bbx_wait / bbx_sync keep their event and pinned error word per THREAD: a lane thread inside _lib.fetch and the caller
    inside ctx.sync() on the same context (round 4 shared one event per context: the second hipEventRecord let the first
    thread's poll return early and fetch read its staging buffer before the copy had landed -- stale values, no error)."""
    import threading
    from blackbox_amd import _lib
    ctx = R.Context(0)
    check = _lib.check
    check(_lib.lib.bbx_set_option(ctx.h, 7, 50), 'bbx_set_option', ctx.h)          # BBX_OPT_WAIT_SLEEP_US
    dev = ctx.device
    stop, bad, rounds = threading.Event(), [], [0]

    def fetcher():
        torch.cuda.set_device(dev)
        s = torch.cuda.Stream(device=dev)
        big = torch.zeros(1 << 26, dtype=torch.float32, device=dev)             # 256 MB: a fill takes ~0.1 ms, several in a row
        small = torch.zeros(64, dtype=torch.float32, device=dev)
        with torch.cuda.stream(s):
            for k in range(1, 151):
                for _ in range(6):
                    big.add_(1.0)                                                # work in front of the value that is fetched
                small.fill_(float(k))
                got = _lib.fetch(ctx, small)
                if not (got == float(k)).all():
                    bad.append((k, got[:4].tolist()))
                rounds[0] = k
        stop.set()

    def syncer():
        torch.cuda.set_device(dev)
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            while not stop.is_set():
                ctx.sync()                                                       # an idle stream: returns at once, recording its event

    ts = [threading.Thread(target=fetcher), threading.Thread(target=syncer)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120.0)
    assert rounds[0] == 150 and not bad, bad[:3]
    ctx.close()
