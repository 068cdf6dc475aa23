"""Throughput path: several frames in flight on one GPU.

The reference farms one frame per worker process (blackbox.py:363-379); on one
MI355X the device work of a frame is ~2-3 ms while the host-side overscan fits
(numpy/scipy, kept bit-compatible with the reference) cost ~15-40 ms of one core.
So a frame goes through three stages and up to [depth] frames overlap:

  A  (GPU, stream A)   overscan strip statistics            -> small D2H (pinned)
  B  (host pool)       16 per-channel fit tasks in worker processes (no GPU there)
  C  (GPU, stream C)   read-noise statistics, fused calibration, mask_init tail,
                       LA-Cosmic, [crosstalk, mask counts, edge fill]

Stage C alternates between [lanes] contexts, each with its own workspace and stream, so
the many short kernels of one frame (flood fill, select, sparse LA-Cosmic steps) overlap
with the streaming kernels of the next; stage A uses no workspace and overlaps with both.
No collective, no inter-GPU traffic: one FramePipeline per GPU / process.
"""
import ctypes as C
import multiprocessing as mp
import os
import queue
import threading
import time

import numpy as np
import torch

from . import _lib, overscan, settings
from . import reduce as R
from ._lib import lib, check

get_par = settings.get_par


def cpu_budget():
    """cores this process may use: the smaller of the cgroup CPU quota (os.cpu_count() reports the whole host inside
    a container) and the affinity mask (taskset / a launcher that pins ranks)"""
    if os.environ.get('BBX_CPU_BUDGET'):                     # (a parent that shares its cores among several processes says so)
        return max(1, int(os.environ['BBX_CPU_BUDGET']))
    try:
        naff = len(os.sched_getaffinity(0))
    except AttributeError:
        naff = os.cpu_count() or 4
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            return max(1, min(naff, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return naff


def default_workers():
    """fit workers per GPU process: the host fits are the CPU-side cost of a frame
    (~7 ms of one core), the orchestrating and lane threads need the rest of the budget for
    their HIP calls (12 of 16 cores measured best on a one-GPU box: 12/13/14 workers gave
    1160/1120/1030 frames/s); BBX_HOST_WORKERS overrides"""
    if 'BBX_HOST_WORKERS' in os.environ:
        return max(1, int(os.environ['BBX_HOST_WORKERS']))
    world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1')))
    return max(2, min(12, cpu_budget() // max(1, world) - 2))


class HostPool:
    """spawned numpy/scipy worker processes for the per-channel overscan fits; they
    never touch the GPU (safe to create after HIP is initialised: spawn, not fork)"""

    def __init__(self, nworkers=None):
        self.n = nworkers or default_workers()
        self.chunk = max(1, int(os.environ.get('BBX_HOST_CHUNK', '4')))
        # one BLAS/OpenMP thread per worker: the fits are tiny, thread pools only fight each other
        saved = {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS')}
        for k in saved:
            os.environ[k] = '1'
        try:
            self.pool = mp.get_context('spawn').Pool(self.n)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        # warm up: import numpy/scipy in every worker
        self.pool.map(_noop, range(self.n * 2))

    def submit(self, fn, tasks):
        # several channels per message: with one task per message the pool's feeder / result
        # threads in this process (pickling, pipe I/O, all under the GIL) become the limit at
        # ~14k tasks/s; BBX_HOST_CHUNK overrides
        return self.pool.map_async(fn, tasks, chunksize=self.chunk)

    def close(self):
        self.pool.close()                   # workers exit after their queue drains (no SIGTERM)
        self.pool.join()


def _noop(i):
    return i


class ShmArena:
    """Per-slot staging buffers in one POSIX shared-memory block, registered with HIP as
    pinned memory: the device copies straight into/out of it and the fit workers map the
    same pages, so a fit task and its result are a few scalars instead of pickled arrays."""

    FIELDS = ('mean', 'hos', 'ninf', 'vfit', 'oscan', 'strip')

    def __init__(self, depth, dy, dx, hos_rows, xsz, name=None):
        from multiprocessing import shared_memory
        self.depth, self.dy, self.dx, self.hos_rows, self.xsz = depth, dy, dx, hos_rows, xsz
        sizes = dict(mean=16 * dy * 8, hos=16 * hos_rows * dx * 4, ninf=8, vfit=16 * dy * 8, oscan=16 * xsz * 8,
                     strip=16 * hos_rows * dx * 4)
        self.off, o = {}, 0
        for k in self.FIELDS:
            self.off[k] = o
            o += (sizes[k] + 4095) // 4096 * 4096
        self.slot_bytes = o
        self.sizes = sizes
        self.owner = name is None
        if self.owner:
            self.shm = shared_memory.SharedMemory(create=True, size=self.slot_bytes * depth)
        else:
            self.shm = shared_memory.SharedMemory(name=name)
        self.name = self.shm.name
        self.registered = False

    def layout(self):
        return (self.name, self.depth, self.dy, self.dx, self.hos_rows, self.xsz)

    def view(self, slot, field):
        shape, dt = {'mean': ((16, self.dy), np.float64), 'hos': ((16, self.hos_rows, self.dx), np.float32),
                     'ninf': ((1,), np.int64),
                     'vfit': ((16, self.dy), np.float64), 'oscan': ((16, self.xsz), np.float64),
                     'strip': ((16, self.hos_rows, self.dx), np.float32)}[field]
        return np.ndarray(shape, dt, buffer=self.shm.buf, offset=slot * self.slot_bytes + self.off[field])

    def span(self, slot, first, last):
        """uint8 view of the slot's bytes from the start of field [first] to the end of field [last]"""
        a = slot * self.slot_bytes + self.off[first]
        n = self.off[last] + self.sizes[last] - self.off[first]
        return np.ndarray((n,), np.uint8, buffer=self.shm.buf, offset=a)

    def register(self):
        """hipHostRegister the block so non_blocking copies are truly asynchronous"""
        addr = C.addressof(C.c_char.from_buffer(self.shm.buf))
        rt = torch.cuda.cudart()
        err = rt.cudaHostRegister(addr, self.shm.size, 0)
        self.registered = int(err) == 0
        self._addr = addr
        return self.registered

    def close(self):
        if self.registered:
            torch.cuda.cudart().cudaHostUnregister(self._addr)
            self.registered = False
        try:
            self.shm.close()
        except BufferError:
            pass                         # numpy views still alive; the mapping goes with the process
        if self.owner:
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass


_ARENAS = {}


def _arena(layout):
    a = _ARENAS.get(layout[0])
    if a is None:
        a = _ARENAS[layout[0]] = ShmArena(*layout[1:], name=layout[0])
    return a


def _shm_solve(args):
    """worker: both phases of channel [c] of the frame staged in [slot] (ML1 path)"""
    layout, slot, c, ysz, xsz, poldeg, tel, data_limit, accum = args
    a = _arena(layout)
    try:
        r = overscan.channel_solve((c, a.view(slot, 'mean')[c], a.view(slot, 'hos')[c], ysz, xsz, poldeg, tel,
                                    data_limit, accum))
    except overscan.OverscanFailure as e:
        return _failed(a, slot, c, e)
    a.view(slot, 'vfit')[c] = r.pop('fit')
    a.view(slot, 'oscan')[c] = r.pop('oscan')
    return r


def _failed(a, slot, c, e):
    """a channel on which os_corr would have raised: its vertical fit is kept (the reference had subtracted it
    already), the horizontal vector is zero; the driver sorts out which channels count (_partial_overscan)"""
    a.view(slot, 'vfit')[c] = e.fit
    a.view(slot, 'oscan')[c] = 0.0
    return dict(failed=str(e), coeffs=e.coeffs, ok=e.ok, level=0.0, dlevel=0.0)


def _shm_phase1(args):
    layout, slot, c, ysz, xsz, poldeg, accum = args
    a = _arena(layout)
    try:
        r = overscan.channel_phase1(c, a.view(slot, 'mean')[c], a.view(slot, 'hos')[c], ysz, xsz, poldeg, accum)
    except overscan.OverscanFailure as e:
        a.view(slot, 'strip')[c] = 0.0
        return _failed(a, slot, c, e)
    a.view(slot, 'vfit')[c] = r.pop('fit')
    a.view(slot, 'strip')[c] = r.pop('strip')
    return r


def _shm_phase2(args):
    layout, slot, c, xsz, tel, data_limit, msr, accum = args
    a = _arena(layout)
    try:
        a.view(slot, 'oscan')[c] = overscan.channel_phase2(c, a.view(slot, 'strip')[c], xsz, tel, data_limit, msr, accum)
    except Exception as e:
        a.view(slot, 'oscan')[c] = 0.0
        return dict(failed='channel {}: horizontal overscan: {}: {}'.format(c + 1, type(e).__name__, e))
    return c


class _Frame:
    __slots__ = ('idx', 'raw', 'header', 'hm', 'state', 'evA', 'h_mean', 'h_hos', 'h_ninf', 'res', 'res2',
                 'evC', 'data', 'mask', 'h_out', 'd_keep', 'p1', 'h_cnt', 'evS', 't0', 'tA', 'tB', 'tC', 'slot', 'lane', 'err',
                 'sub', 'failed', 'os_ok', 'os_fail', 'out_group', 'out_base', 'out_names', 'written_name', 'staged_wanted')


class _LaneCtx:
    """a library context bound to one stream (what the stage functions of reduce.py ask of a
    Context: .h, .device, .stream(), .sync()) -- the stream handle is looked up once, not per call"""

    def __init__(self, ctx, stream):
        self.h, self.device, self.owner = ctx.h, ctx.device, ctx
        self.torch_stream = stream
        self.sp = C.c_void_p(stream.cuda_stream)

    def stream(self):
        return self.sp

    def sync(self):
        check(lib.bbx_sync(self.h, self.sp), 'bbx_sync', self.h)


class _LaneThread(threading.Thread):
    """issues the device stage of one lane: the launches of a frame take ~0.6 ms of host time,
    most of it inside the library (ctypes drops the GIL there), so the lanes and the
    orchestrating thread overlap.  Work items: (method, frame, results)."""

    _profiled = False

    def __init__(self, pipe, device, name='bbx-lane'):
        super().__init__(daemon=True, name=name)
        self.pipe, self.device, self.q = pipe, device, queue.SimpleQueue()

    def run(self):
        torch.cuda.set_device(self.device)              # the current HIP device is per thread
        prof = None
        if os.environ.get('BBX_LANE_PROFILE') and not _LaneThread._profiled:      # (debug: cProfile of the first lane thread)
            import cProfile
            _LaneThread._profiled = True
            prof = cProfile.Profile()
            prof.enable()
        while True:
            item = self.q.get()
            if item is None:
                if prof is not None:
                    import pstats
                    prof.disable()
                    with open(os.environ['BBX_LANE_PROFILE'], 'w') as fh:
                        st = pstats.Stats(prof, stream=fh)
                        st.sort_stats('tottime').print_stats(45)
                        st.sort_stats('cumulative').print_stats(60)
                return
            fn, f, results = item
            try:
                fn(f, results)
            except BaseException as e:                  # handed to the orchestrating thread
                f.err = e
                f.state = 'err'


def _new_event():
    ev = C.c_void_p()
    check(lib.bbx_event_create(C.byref(ev)), 'bbx_event_create')
    return ev


class FramePipeline:
    def __init__(self, ctx, tel, geom, mflat=None, mbias=None, bpm=None, xtalk_coeffs=None, exptime=60.0,
                 pool=None, depth=4, do_cosmics=True, do_finish=False, accum='f32seq', keep_outputs=False, lanes=2,
                 detect_sats=False, subtract=None, log=None, outstage=None, out_base=None, on_written=None, header_hook=None,
                 stage_limmag=True):
        """do_finish: crosstalk, mask counts, edge fill (the tail of blackbox_reduce);
        detect_sats: satellite trails before them (blackbox.py:1919-1952);
        subtract: dict of keyword arguments for zogy.optimal_subtraction (ref=, ref_mask=, psf_new=,
        psf_ref=, ...) -- the frame then continues into the background mesh / ZOGY / photometry
        stage on the same lane (blackbox.py:2350-2354, 2460-2465), results in frame.sub;
        outstage: an outstage.OutputStage -- the frame's image products (_red, _mask, and with a subtraction _D, _Scorr,
        _Fpsf, _trans_limmag) are tile-compressed on the lane right behind the kernels that made them and written as
        `.fits.fz` by the stage's writer threads (blackbox.py:1981-1990, 812-857); out_base(idx, header) -> path of the
        frame's _red product without '.fits'; on_written(frame, group) is called by a writer thread when the last file
        of the frame is on disk; header_hook(frame, headers) -> headers lets the caller complete the headers of the
        frame's files ({path: header, None: default}) when the frame's scalars are in, before the writers use them"""
        self.ctx, self.tel, self.geom = ctx, tel, geom
        # stage-C lanes: (context, stream); lane 0 is the caller's context
        self.own_ctx = [R.Context(ctx.device.index) for _ in range(max(1, lanes) - 1)]
        self.lane_stream = [torch.cuda.Stream(device=ctx.device) for _ in range(max(1, lanes))]
        self.lane_ctx = [_LaneCtx(c, st) for c, st in zip([ctx] + self.own_ctx, self.lane_stream)]
        # many host threads wait on this GPU (lanes, readers, writers): their waits poll + sleep instead of spinning on a core
        # each (include/bbx.h, BBX_OPT_WAIT_SLEEP_US; BBX_WAIT_SLEEP_US=0 keeps the runtime's spinning waits); the caller's
        # context gets its setting back in close()
        for c in self.own_ctx:
            check(lib.bbx_set_option(c.h, 7, _lib.WAIT_SLEEP_US), 'bbx_set_option', c.h)
        check(lib.bbx_set_option(ctx.h, 7, _lib.WAIT_SLEEP_US), 'bbx_set_option', ctx.h)
        self.mflat, self.bpm = mflat, bpm
        self.mbias = mbias if (mbias is not None and get_par(settings.subtract_mbias, tel)) else None
        self.xtalk = xtalk_coeffs
        self.exptime = exptime
        self.pool = pool or HostPool()
        self.own_pool = pool is None
        self.depth = depth
        self.do_cosmics, self.do_finish, self.accum = do_cosmics, do_finish, accum
        self.detect_sats = detect_sats and bool(get_par(settings.detect_sats, tel))
        self.subtract = dict(subtract) if subtract else None
        # one lane at a time inside bbx_zogy_frame (BBX_ZOGY_GATE=0 switches the gate off)
        self.ref_bkg_std = None
        self.zogy_gate = None
        self.fpack_in_gate = os.environ.get('BBX_FPACK_GATE', '0') == '1'
        if self.subtract and os.environ.get('BBX_ZOGY_GATE', '1') != '0':
            from . import zogy as G
            # (BBX_ZOGY_PRIO=1: the sections on one high-priority stream instead of the lanes' own -- measured in round 3: the
            # section's kernels take 13.5 instead of 7.9 ms per frame in the pipeline and the frame rate drops 2 %)
            self.zogy_gate = G.StreamGate(device=ctx.device, priority=os.environ.get('BBX_ZOGY_PRIO', '0') == '1')
        self.keep_sub = ('D', 'Scorr', 'Fpsf', 'Fpsferr')        # device products kept on the frame when keep_outputs
        self.log = log
        self.outstage, self.out_base, self.on_written, self.header_hook = outstage, out_base, on_written, header_hook
        # False: the caller makes `_trans_limmag` itself (in magnitudes, with a zeropoint); a callable(header) decides per frame
        self.stage_limmag = stage_limmag
        if outstage is not None and out_base is None:
            raise ValueError('outstage needs out_base')
        self.keep_outputs = keep_outputs
        self.sA = torch.cuda.Stream(device=ctx.device)
        # stage A has a library context of its own: it is driven from the orchestrating thread
        # while lane 0 (the caller's context) is driven from a lane thread
        self.own_ctx.append(R.Context(ctx.device.index))
        self.ctxA = _LaneCtx(self.own_ctx[-1], self.sA)
        self._finisher = None
        # A dozen threads of this process take turns on the interpreter lock: lane threads and the orchestrating thread make
        # many short library calls (each gives the lock up and must get it back), the completion hook and the header formatting
        # of the writers are plain Python.  At CPython's default switch interval (5 ms) a thread that wants the lock back waits
        # up to that long per call while a Python-bound thread runs: a convoy (measured: the list run of blackbox.py fell from
        # 37 to 13 frames/s when the hook moved to a thread of its own).  0.2 ms keeps the hand-over latency below a launch.
        import sys
        us = float(os.environ.get('BBX_SWITCH_US', '200'))
        if us > 0 and sys.getswitchinterval() > us * 1e-6:
            sys.setswitchinterval(us * 1e-6)
        self.lane_thread = [_LaneThread(self, ctx.device) for _ in self.lane_ctx]
        for t in self.lane_thread:
            t.start()
        # outputs: one (data, mask) pair per lane unless the caller keeps them (frames of a lane
        # are ordered on its stream, so the pair is free again when the next one starts)
        ny, nx = 2 * geom.ysize_chan, 8 * geom.xsize_chan
        self.lane_out = None if (keep_outputs or outstage is not None) else [
            (torch.empty((ny, nx), dtype=torch.float32, device=ctx.device),
             torch.empty((ny, nx), dtype=torch.uint8, device=ctx.device)) for _ in self.lane_ctx]
        self.gain = get_par(settings.gain, tel)
        self.g32 = _lib.f32x16(self.gain)
        g = geom
        self.dy, self.dx = g.ny_raw // 2, g.nx_raw // 8
        self.ysz, self.xsz = g.ysize_chan, g.xsize_chan
        self.hos_rows = self.dy - self.ysz - 10
        self.two_phase = tel != 'ML1'
        self.t_stats = [0.0, 0.0, 0.0, 0]      # wall: start->stats on host, fits, device stage; frames
        self.lane_cpu = [0.0, 0.0, 0]          # lane threads: CPU seconds, wall seconds inside the device stage, frames
        # staging buffers, one set per frame in flight (pinned host + device), allocated once:
        # pin_memory()/hipHostMalloc costs ~10 ms per call and must stay out of the frame loop
        dev = ctx.device
        self.arena = ShmArena(depth, self.dy, self.dx, self.hos_rows, self.xsz)
        if not self.arena.register():
            raise RuntimeError('hipHostRegister of the staging arena failed')
        self.layout = self.arena.layout()
        # per slot: the device mirrors of the arena regions (one copy each way per stage instead
        # of one per array: a torch copy_ costs ~30 us of the orchestrating thread) and one
        # packed result record
        ar = self.arena
        self.slots = []
        for i in range(depth):
            n_in = ar.off['ninf'] + ar.sizes['ninf'] - ar.off['mean']
            n_vo = ar.off['oscan'] + ar.sizes['oscan'] - ar.off['vfit']
            d_in = torch.empty(n_in, dtype=torch.uint8, device=dev)
            d_vo = torch.empty(n_vo, dtype=torch.uint8, device=dev)

            def dv(buf, base, k, dt, shape=None):
                o = ar.off[k] - ar.off[base]
                t = buf[o:o + ar.sizes[k]].view(dt)
                return t.view(shape) if shape else t
            # std[16] f64 | nobj i32 | stats[16] i32 | nsats i32 | cnt[6] i64 | step error flags [8] i32 (248..280)
            d_res = torch.zeros(288, dtype=torch.uint8, device=dev)
            h_res = torch.zeros(288, dtype=torch.uint8, pin_memory=True)
            h_in_np, h_vo_np = ar.span(i, 'mean', 'ninf'), ar.span(i, 'vfit', 'oscan')
            h_cnt = torch.empty((2, 16, self.xsz), dtype=torch.int32, pin_memory=True)
            d_cnt = torch.empty((2, 16, self.xsz), dtype=torch.int32, device=dev)
            self.slots.append(dict(
                evA=_new_event(), evS=_new_event(), evC=_new_event(),
                # (dst, src, nbytes, kind) of the packed copies, for bbx_copy_async
                cp_in=(C.c_void_p(h_in_np.ctypes.data), R._ptr(d_in), n_in, 1),
                cp_vo=(R._ptr(d_vo), C.c_void_p(h_vo_np.ctypes.data), n_vo, 0),
                cp_res=(R._ptr(h_res), R._ptr(d_res), 288, 1),
                cp_cnt=(R._ptr(h_cnt), R._ptr(d_cnt), h_cnt.numel() * 4, 1),
                d_in=d_in, h_in=torch.from_numpy(ar.span(i, 'mean', 'ninf')),
                d_mean=dv(d_in, 'mean', 'mean', torch.float64),
                d_hos=dv(d_in, 'mean', 'hos', torch.float32, (16, self.hos_rows, self.dx)),
                d_ninf=dv(d_in, 'mean', 'ninf', torch.int64),
                h_ninf=torch.from_numpy(ar.view(i, 'ninf')),
                d_vo=d_vo, h_vo=torch.from_numpy(ar.span(i, 'vfit', 'oscan')),
                d_vfit=dv(d_vo, 'vfit', 'vfit', torch.float64), d_oscan=dv(d_vo, 'vfit', 'oscan', torch.float64),
                d_res=d_res, h_res=h_res,
                d_std=d_res[0:128].view(torch.float64), d_nobj=d_res[128:132].view(torch.int32),
                d_stats=d_res[132:196].view(torch.int32), d_nsats=d_res[196:200].view(torch.int32),
                d_cnt6=d_res[200:248].view(torch.int64), d_steps=d_res[248:280].view(torch.int32),
                d_info=torch.zeros(8, dtype=torch.float32, device=dev), d_med=torch.empty(16, dtype=torch.float32, device=dev),
                h_std=h_res[0:128].view(torch.float64), h_nobj=h_res[128:132].view(torch.int32),
                h_stats=h_res[132:196].view(torch.int32), h_nsats=h_res[196:200].view(torch.int32),
                h_cnt6=h_res[200:248].view(torch.int64), h_steps=h_res[248:280].view(torch.int32),
                h_cnt=h_cnt, d_cnt=d_cnt))
        # header keys / comments are the same for every frame
        self._k_bias = [[('BIAS{}A{}'.format(c + 1, k), '[e-] channel {} vert. overscan A{} polyfit coeff'.format(c + 1, k))
                         for k in range(settings.voscan_poldeg + 1)] for c in range(16)]
        self._k_vfitok = [('VFITOK{}'.format(c + 1), 'channel {} vert. overscan polyfit finite?'.format(c + 1)) for c in range(16)]
        self._k_biasm = [('BIASM{}'.format(c + 1), '[e-] channel {} mean vertical overscan'.format(c + 1)) for c in range(16)]
        self._k_rdn = [('RDN{}'.format(c + 1), '[e-] channel {} sigma (STD) vertical overscan'.format(c + 1)) for c in range(16)]
        self.free_slots = list(range(depth))
        # frames still to run with LA-Cosmic's background level prepared in advance (include/bbx.h,
        # BBX_OPT_LAC_LEVEL_FEED); BBX_LAC_FEED_FRAMES sets the start value (0: only after a frame needed it)
        self.level_feed_left = int(os.environ.get('BBX_LAC_FEED_FRAMES', '0'))

    def close(self):
        lib.bbx_set_option(self.ctx.h, 7, 0)
        if self.own_pool:
            self.pool.close()
        for t in self.lane_thread + ([self._finisher] if self._finisher is not None else []):
            t.q.put(None)
        for t in self.lane_thread + ([self._finisher] if self._finisher is not None else []):
            t.join()
        self.lane_thread, self._finisher = [], None
        for sl in self.slots:
            for k in ('evA', 'evS', 'evC'):
                lib.bbx_event_destroy(sl[k])
        self.slots = []
        self.lane_out = None
        self.arena.close()
        for c in self.own_ctx:
            c.close()

    # ---- stage A ------------------------------------------------------------------
    def _start(self, idx, raw, header, ready=None):
        """ready: a torch.cuda.Event after which [raw] holds the frame (instage.InputStage: the decode runs on a reader's
        stream); stage A's stream waits for it, and every later stage of the frame waits for stage A"""
        ctx = self.ctxA
        # device pointers cross the C ABI without their extents: refuse a frame of another shape here
        if (not torch.is_tensor(raw) or not raw.is_cuda or not raw.is_contiguous()
                or tuple(raw.shape) != (self.geom.ny_raw, self.geom.nx_raw)):
            raise ValueError('frame {}: contiguous device tensor of shape {} expected'.format(
                idx, (self.geom.ny_raw, self.geom.nx_raw)))
        f = _Frame()
        f.idx, f.raw, f.header, f.hm, f.t0, f.err = idx, raw, header, {}, time.perf_counter(), None
        R.gain_corr(header, self.tel)
        f.slot = self.free_slots.pop()
        f.lane = idx % len(self.lane_ctx)
        sl = self.slots[f.slot]
        sA = self.ctxA.sp
        if ready is not None:
            self.sA.wait_event(ready)
        d_mean, d_hos, d_ninf = sl['d_mean'], sl['d_hos'], sl['d_ninf']
        check(lib.bbx_overscan_stats(ctx.h, C.byref(self.geom), R._ptr(raw), R.raw_type_of(raw), self.g32,
                                     R._ptr(d_mean), R._ptr(d_hos), R._ptr(d_ninf), sA),
              'bbx_overscan_stats', ctx.h)
        f.h_ninf = sl['h_ninf']
        check(lib.bbx_copy_async(*sl['cp_in'], sA), 'bbx_copy_async')     # mean | hos | ninf in one copy
        f.evA = sl['evA']
        check(lib.bbx_event_record(f.evA, sA), 'bbx_event_record')
        f.state = 'A'
        return f

    # ---- stage B ------------------------------------------------------------------
    def _submit_fits(self, f):
        if not self.two_phase:
            tasks = [(self.layout, f.slot, c, self.ysz, self.xsz, settings.voscan_poldeg, self.tel, 2000, self.accum)
                     for c in range(16)]
            f.res = self.pool.submit(_shm_solve, tasks)
        else:
            tasks = [(self.layout, f.slot, c, self.ysz, self.xsz, settings.voscan_poldeg, self.accum)
                     for c in range(16)]
            f.res = self.pool.submit(_shm_phase1, tasks)
        f.d_keep = None
        f.state = 'B'

    def _fill_header_vos(self, f, results):
        h = f.header
        h['N-INFNAN'] = (int(f.h_ninf.item()), 'number of pixels with infinite/nan values')
        for c, r in enumerate(results):
            for (key, comment), v in zip(self._k_bias[c], r['coeffs']):
                h[key] = (float(v) if np.isfinite(v) else 'None', comment)
            h[self._k_vfitok[c][0]] = (bool(r['ok']), self._k_vfitok[c][1])
        mean_vos = np.array([r['level'] for r in results])
        for c in range(16):
            h[self._k_biasm[c][0]] = (float(mean_vos[c]), self._k_biasm[c][1])
        h['BIASMEAN'] = (float(np.nanmean(mean_vos)), '[e-] average all channel means vert. overscan')

    def _satcol(self, f, results):
        """BlackGEM: per-column saturation counts need the vertical fit (two-phase)"""
        ctx, dev = self.lane_ctx[f.lane], self.ctx.device
        f.p1 = results
        lim = settings.os_ypix_lim[self.tel]
        satl = np.array(get_par(settings.satlevel, self.tel)) * np.array(self.gain)
        sp = ctx.sp
        check(lib.bbx_stream_wait_event(sp, f.evA), 'bbx_stream_wait_event')
        sl = self.slots[f.slot]
        d_vfit = sl['d_vfit']
        check(lib.bbx_copy_async(*sl['cp_vo'], sp), 'bbx_copy_async')     # vfit (| oscan, not final yet)
        d_cnt = sl['d_cnt']
        check(lib.bbx_satcol_counts(ctx.h, C.byref(self.geom), R._ptr(f.raw), R.raw_type_of(f.raw), self.g32,
                                    R._ptr(d_vfit), _lib.f32x16(np.float32(0.9 * satl)), int(lim[0]), int(lim[1]),
                                    R._ptr(d_cnt), sp), 'bbx_satcol_counts', ctx.h)
        f.h_cnt = sl['h_cnt']
        check(lib.bbx_copy_async(*sl['cp_cnt'], sp), 'bbx_copy_async')
        f.evS = sl['evS']
        check(lib.bbx_event_record(f.evS, sp), 'bbx_event_record')
        f.d_keep = (d_vfit, d_cnt)
        f.state = 'S'

    def _submit_phase2(self, f):
        cnt = f.h_cnt.numpy()
        msr = (cnt[0] >= 3) | (cnt[1] >= 10)
        tasks = [(self.layout, f.slot, c, self.xsz, self.tel, 2000, msr[c], self.accum) for c in range(16)]
        f.res2 = self.pool.submit(_shm_phase2, tasks)
        f.state = 'B2'

    def _partial_overscan(self, f, results):
        """channels on which os_corr would have raised (overscan.OverscanFailure): the reference stops at the first
        of them and crops its half-processed array -- keep the vectors of the channels before it and the vertical
        fit of that one, zero the rest (the arena holds the vectors the device stage uploads)"""
        f.os_fail = None
        if results is None:
            return None
        bad = [c for c, r in enumerate(results) if r.get('failed')]
        if bad:
            k = bad[0]
            f.os_fail = (k, results[k]['failed'])
            self.arena.view(f.slot, 'vfit')[k + 1:] = 0.0
            self.arena.view(f.slot, 'oscan')[k:] = 0.0
            if self.log is not None:
                self.log.error('frame %d: os_corr failed (%s); adopting an overscan of zero for all channels', f.idx,
                               results[k]['failed'])
        return results

    # ---- stage C ------------------------------------------------------------------
    def _device_stage(self, f, results):
        ctx = self.lane_ctx[f.lane]
        # every torch allocation / operation of the stage belongs to the lane's stream (the caching
        # allocator hands a freed block only to work queued behind it on the same stream)
        t_cpu, t_wall = time.thread_time(), time.perf_counter()
        with torch.cuda.stream(ctx.torch_stream):
            self._device_stage_on_lane(f, results, ctx)
        # host cost of the stage: CPU seconds of this lane thread (what it holds the interpreter for, roughly) and its wall time
        self.lane_cpu[0] += time.thread_time() - t_cpu
        self.lane_cpu[1] += time.perf_counter() - t_wall
        self.lane_cpu[2] += 1

    def _device_stage_on_lane(self, f, results, ctx):
        geom, tel = self.geom, self.tel
        sp = ctx.sp
        h, hm = f.header, f.hm
        sl = self.slots[f.slot]
        d_steps = sl['d_steps']
        f.failed, f.sub = [], None
        check(lib.bbx_stream_wait_event(sp, f.evA), 'bbx_stream_wait_event')
        sol = R.OverscanSolution()
        d_std = sl['d_std']
        if results is None:
            # the host fits failed: overscan of zero, RDN = 10 (blackbox.py:1537-1585)
            zs = R.zero_overscan_solution(ctx, h, geom)
            sol.vfit, sol.oscan, sol.d_vfit, sol.d_oscan = zs.vfit, zs.oscan, zs.d_vfit, zs.d_oscan
            h['N-INFNAN'] = (int(f.h_ninf.item()), 'number of pixels with infinite/nan values')
            d_std.fill_(10.0)
            f.os_ok = False
        elif getattr(f, 'os_fail', None) is not None:
            # os_corr raised part-way: the crop of the reference's half-processed array (_partial_overscan)
            k = f.os_fail[0]
            for c, r in enumerate(results[:k + 1]):
                if r.get('coeffs') is not None:
                    for (key, comment), v in zip(self._k_bias[c], r['coeffs']):
                        h[key] = (float(v) if np.isfinite(v) else 'None', comment)
                    h[self._k_vfitok[c][0]] = (bool(r['ok']), self._k_vfitok[c][1])
            R.zero_overscan_solution(None, h, geom, header_only=True)
            h['N-INFNAN'] = (int(f.h_ninf.item()), 'number of pixels with infinite/nan values')
            sol.vfit, sol.oscan = self.arena.view(f.slot, 'vfit'), self.arena.view(f.slot, 'oscan')
            sol.d_vfit, sol.d_oscan = sl['d_vfit'], sl['d_oscan']
            check(lib.bbx_copy_async(*sl['cp_vo'], sp), 'bbx_copy_async')
            d_std.fill_(10.0)
            f.os_ok = False
        else:
            self._fill_header_vos(f, results)
            dlevel = np.float32([r['dlevel'] for r in results])
            sol.vfit, sol.oscan = self.arena.view(f.slot, 'vfit'), self.arena.view(f.slot, 'oscan')
            sol.d_vfit, sol.d_oscan = sl['d_vfit'], sl['d_oscan']
            check(lib.bbx_copy_async(*sl['cp_vo'], sp), 'bbx_copy_async')     # vfit | oscan in one copy
            check(lib.bbx_vos_std(ctx.h, C.byref(geom), R._ptr(f.raw), R.raw_type_of(f.raw), self.g32,
                                  R._ptr(sol.d_vfit), _lib.f32x16(dlevel), R._ptr(d_std), sp),
                  'bbx_vos_std', ctx.h)
            f.os_ok = True
        h['OS-P'] = (f.os_ok, 'corrected for overscan?')
        out = self.lane_out[f.lane] if self.lane_out is not None else None
        data, mask = R.calibrate(ctx, f.raw, sol, h, hm, tel, geom, mbias=self.mbias, mflat=self.mflat, bpm=self.bpm, out=out)
        h['MBIAS-P'] = (self.mbias is not None, 'corrected for master bias?')
        h['MFLAT-P'] = (self.mflat is not None, 'corrected for master flat?')
        R.step_mark(ctx, d_steps, 'calibrate')
        try:
            d_nobj = R.mask_init_finish(ctx, mask, h, hm, geom, d_n=sl['d_nobj'])
            h['MASK-P'] = (True, 'mask image created?')
        except _lib.BBXError:
            d_nobj = None
            h['MASK-P'] = (False, 'mask image created?')
        R.step_mark(ctx, d_steps, 'mask')
        d_stats = None
        if self.do_cosmics:
            try:
                # background level: prepared in advance only while recent frames needed it (st[15])
                check(lib.bbx_set_option(ctx.h, 1, 1 if self.level_feed_left > 0 else 0), 'bbx_set_option', ctx.h)
                # RDNOISE = nanmean of the 16 channel sigmas is formed on the device
                d_stats = R.cosmics_corr(ctx, data, h, mask, hm, tel, d_rdn16=d_std, d_stats=sl['d_stats'])
                h['COSMIC-P'] = (True, 'corrected for cosmic rays?')
            except _lib.BBXError:
                d_stats = None
                f.failed.append('cosmics')
        R.step_mark(ctx, d_steps, 'cosmics')
        d_cnt = sl['d_cnt6']
        if self.do_finish and self.xtalk is not None:
            try:
                R.xtalk_corr(ctx, data, self.xtalk, mask, geom)
                h['XTALK-P'] = (True, 'corrected for crosstalk?')
            except _lib.BBXError:
                h['XTALK-P'] = (False, 'corrected for crosstalk?')
        R.step_mark(ctx, d_steps, 'xtalk')
        d_nsats = None
        if self.detect_sats:
            try:
                d_nsats = sl['d_nsats']
                check(lib.bbx_sat_trails(ctx.h, data.shape[0], data.shape[1], R._ptr(data), R._ptr(mask), R.sat_cos_sin(),
                                         R.NTHETA_SAT, *R.sat_gauss_weights(), R._ptr(d_nsats), R._ptr(sl['d_info']), sp),
                      'bbx_sat_trails', ctx.h)
                h['SAT-P'] = (True, 'processed for satellite trails?')
            except _lib.BBXError:
                d_nsats = None
                f.failed.append('sat')
        R.step_mark(ctx, d_steps, 'sat')
        if self.do_finish:
            check(lib.bbx_mask_counts(ctx.h, mask.numel(), R._ptr(mask), R._ptr(d_cnt), sp), 'bbx_mask_counts', ctx.h)
            check(lib.bbx_edge_fill(ctx.h, C.byref(geom), R._ptr(data), R._ptr(mask), R._ptr(sl['d_med']), sp),
                  'bbx_edge_fill', ctx.h)
        R.step_mark(ctx, d_steps, 'finish')
        if self.subtract is not None:
            from . import zogy as G
            try:
                sub = G.optimal_subtraction(ctx, data, new_mask=mask, zogy_gate=self.zogy_gate, ref_bkg_std=self.ref_bkg_std,
                                            **self.subtract)
                if (self.ref_bkg_std is None and self.subtract.get('ref_is_bkgsub') and self.subtract.get('ref_bkg_std_mini') is not None
                        and self.subtract.get('ref_grid') is None and 'bkg_std_ref' in sub):
                    # the reference's sigma image is the same for every frame of the run: made once, on whichever lane
                    # comes first (its stream has finished with it before any other lane can pick it up: see below)
                    check(lib.bbx_wait(ctx.h, ctx.sp), 'bbx_wait', ctx.h)
                    self.ref_bkg_std = sub['bkg_std_ref']
                f.sub = sub
            except (_lib.BBXError, ValueError) as e:
                f.failed.append('zogy')
                if self.log is not None:
                    self.log.error('frame %d: [optimal_subtraction] failed: %s', f.idx, e)
            R.step_mark(ctx, d_steps, 'zogy')
        f.out_group = None
        if self.outstage is not None:
            if self.zogy_gate is not None and self.fpack_in_gate:
                # the compression of the frame's images as a section of its own behind the subtraction's: `k_fp_tile` and the
                # ZOGY kernels each hold a CU's LDS and wave slots in pairs; side by side each runs with half of them
                with self.zogy_gate:
                    self._submit_outputs(f, ctx, data, mask)
            else:
                self._submit_outputs(f, ctx, data, mask)
        if not self.keep_outputs and f.sub is not None:
            for k in list(f.sub):
                if torch.is_tensor(f.sub[k]):
                    del f.sub[k]                                      # device products stay only on request
        # scalar results: one small pinned D2H of the packed record
        f.h_out = (sl['h_std'], sl['h_nobj'] if d_nobj is not None else None, sl['h_stats'] if d_stats is not None else None,
                   sl['h_cnt6'], sl['h_nsats'] if d_nsats is not None else None, sl['h_steps'])
        # (by a kernel: the record must not queue behind the output stage's transfers on the copy engines)
        check(lib.bbx_copy_kernel(sl['cp_res'][0], sl['cp_res'][1], sl['cp_res'][2], sp), 'bbx_copy_kernel')
        f.evC = sl['evC']
        check(lib.bbx_event_record(f.evC, sp), 'bbx_event_record')
        f.d_keep = (sol, d_std, d_nobj, d_stats, d_cnt)
        f.data, f.mask = (data, mask) if self.keep_outputs else (None, None)
        f.state = 'C'

    def _submit_outputs(self, f, ctx, data, mask):
        """queue the frame's image products on this lane for the output stage (no host wait)"""
        st = self.outstage
        base = self.out_base(f.idx, f.header)
        g = st.new_group(f, self._group_done)
        f.out_group, f.out_base = g, base
        names = {}
        names['red'] = st.submit(ctx, g, data, base + '.fits')
        names['mask'] = st.submit(ctx, g, mask, base.replace('_red', '_mask') + '.fits') if base.endswith('_red') else \
            st.submit(ctx, g, mask, base + '_mask.fits')
        sub = f.sub
        if sub is not None and sub.get('D') is not None:
            for ext in ('D', 'Scorr', 'Fpsf'):
                names[ext] = st.submit(ctx, g, sub[ext], '{}_{}.fits'.format(base, ext))
            stage_lim = self.stage_limmag(f.header) if callable(self.stage_limmag) else self.stage_limmag
            if stage_lim:
                nsig = float(sub['header_trans']['T-NSIGMA'][0])
                # `_trans_limmag` as a flux limit (no zeropoint on this path): T-NSIGMA x Fpsferr, multiplied by the compression
                # kernel as it loads Fpsferr (no image of its own)
                names['limmag'] = st.submit(ctx, g, sub['Fpsferr'], base + '_trans_limmag.fits', scale=nsig)
        f.out_names = names
        g.seal()

    def _group_done(self, g):
        if self.on_written is not None:
            self.on_written(g.token, g)

    def _finalize(self, f):
        h, hm = f.header, f.hm
        if f.os_ok:
            std = f.h_out[0].numpy()
            for c in range(16):
                h[self._k_rdn[c][0]] = (float(std[c]), self._k_rdn[c][1])
            h['RDNOISE'] = (float(np.nanmean(std)), '[e-] average all channel sigmas vert. overscan')
        if f.h_out[1] is not None:
            nobj = int(f.h_out[1].item())
            h['NOBJ-SAT'] = hm['NOBJ-SAT'] = (nobj, 'number of saturated objects')
        if f.h_out[2] is not None:
            st = f.h_out[2].numpy()
            h['NCOSMICS'] = hm['NCOSMICS'] = (float(st[6]) / float(self.exptime), '[/s] number of cosmic rays identified')
            h['NCRPIX'] = (int(st[7]), 'number of cosmic-ray pixels')
            if st[15]:
                self.level_feed_left = 64             # a frame needed the level: feed it for the next frames
            elif self.level_feed_left > 0:
                self.level_feed_left -= 1
        if f.h_out[4] is not None:
            h['NSATS'] = hm['NSATS'] = (int(f.h_out[4].item()), 'number of satellite trails identified')
        if self.do_finish:
            R.fill_mask_header(hm, f.h_out[3].numpy())
        for step in f.failed:                                    # host-side failures of a step's launch
            if step == 'cosmics':
                h['COSMIC-P'] = (False, 'corrected for cosmic rays?')
                h['NCOSMICS'] = hm['NCOSMICS'] = ('None', '[/s] number of cosmic rays identified')
            elif step == 'sat':
                h['SAT-P'] = (False, 'processed for satellite trails?')
                h['NSATS'] = hm['NSATS'] = ('None', 'number of satellite trails identified')
        # device-side error flags per step (bbx_step_mark) -> <STEP>-P False for this frame only
        f.failed += R.apply_step_errors(h, hm, f.h_out[5].numpy(), self.log)
        if getattr(f, 'out_group', None) is not None:
            # the headers the writers were waiting for: reduction keywords for _red, mask keywords for _mask, reduction +
            # transient keywords for the subtraction images
            hdrs = {None: h, f.out_names['mask']: hm}
            if f.sub is not None and 'header_new' in f.sub:
                h.update(f.sub['header_new'])
                ht = dict(h); ht.update(f.sub.get('header_trans', {}))
                for k in ('D', 'Scorr', 'Fpsf', 'limmag'):
                    if k in f.out_names:
                        hdrs[f.out_names[k]] = ht
            if self.header_hook is not None:
                # the caller's completion of the frame (header bookkeeping, QC flags, its small files: ~15 ms of Python per
                # frame in blackbox.py) runs on a thread of its own: the orchestrating thread goes back to its frames at once;
                # the frame counts as done -- on_done, its slot -- when the hook has returned
                f.d_keep = None
                f.state = 'H'
                if self._finisher is None:
                    self._finisher = _LaneThread(self, self.ctx.device, name='bbx-finisher')
                    self._finisher.start()
                self._finisher.q.put((self._run_hook, f, hdrs))
                return
            f.out_group.set_headers(hdrs)
        f.d_keep = None
        f.state = 'done'

    def _run_hook(self, f, hdrs):
        try:
            hdrs = self.header_hook(f, hdrs)
        except BaseException as e:                                  # the files still get the default headers
            if self.log is not None:
                self.log.exception('frame %d: header hook failed: %s', f.idx, e)
        f.out_group.set_headers(hdrs)
        f.state = 'done'

    # ---- driver --------------------------------------------------------------------
    def run(self, frames, on_done=None, on_input_error=None):
        """frames: iterable of (raw device tensor, header dict[, ready event]).  Processes all of them with
        up to [depth] in flight; calls on_done(idx, frame) in completion order.
        on_input_error(idx, exc): a source that fails ONE frame raises instage.InputError in its place and goes on with the
        next (instage.InputStage does; a generator cannot): the frame's index is skipped, the hook is told, the run goes on
        (blackbox.py:948-999: the reference fails that file and continues).  Without the hook the error ends the run."""
        it = iter(frames)
        self._has_next = getattr(frames, 'has_next', None)        # (instage.InputStage: ask before taking a frame)
        self._on_input_error = on_input_error
        live, ndone, exhausted = [], 0, False
        try:
            return self._run(it, live, ndone, exhausted, on_done)
        except BaseException as e:
            for f in live:
                if getattr(f, 'err', None) is None:
                    f.err = e
            self._abort(live)
            raise

    def _abort(self, live):
        """after an error: let the fit workers and the lanes finish what was queued, drain the streams
        and take all slots back"""
        for f in live:
            for r in (getattr(f, 'res', None), getattr(f, 'res2', None)):
                try:
                    if r is not None:
                        r.wait(30.0)
                except Exception:
                    pass
        threads = self.lane_thread + ([self._finisher] if self._finisher is not None else [])
        done = [threading.Event() for _ in threads]
        for t, ev in zip(threads, done):
            t.q.put((lambda f, r, ev=ev: ev.set(), None, None))
        for ev in done:
            ev.wait(30.0)
        try:
            torch.cuda.synchronize(self.ctx.device)
        except Exception:
            pass
        # frames of the output stage that will never be finalised: their writers wait for headers -- tell them
        # (the lanes have drained: every live frame that got as far as _submit_outputs has its group by now)
        for f in live:
            g = getattr(f, 'out_group', None)
            if g is not None and not g.header_ready.is_set():
                g.cancel(getattr(f, 'err', None))
        self.free_slots = list(range(self.depth))

    def _run(self, it, live, ndone, exhausted, on_done):
        from .instage import InputError
        nxt = 0                                                    # index of the next frame of the source (failed ones count)
        while True:
            progressed = False
            while not exhausted and len(live) < self.depth:
                if self._has_next is not None and not self._has_next():
                    break                                          # the next frame is still being read / decoded
                try:
                    item = next(it)
                except StopIteration:
                    exhausted = True
                    break
                except InputError as e:
                    if self._on_input_error is None:
                        raise
                    if self.log is not None:
                        self.log.error('frame %d: input failed: %s', nxt, e)
                    self._on_input_error(nxt, e)
                    nxt += 1
                    progressed = True
                    continue
                idx = nxt
                nxt += 1
                raw, header = item[0], item[1]
                live.append(self._start(idx, raw, header, item[2] if len(item) > 2 else None))
                progressed = True
            # events of one stream complete in order: behind the first frame whose stage-A event (one stream for all frames) or
            # whose lane's event is still pending, the later frames of that stream need no query of their own -- with 48
            # frames in flight the loop made ~40 library calls per round, each giving the interpreter lock away
            a_pending, lane_pending = False, set()
            for f in live:
                st = f.state                                       # (read once: lane threads move a frame from 'Q' to 'S' / 'C' meanwhile)
                if st == 'A':
                    if a_pending:
                        continue
                    if lib.bbx_event_query(f.evA) != 1:
                        a_pending = True
                        continue
                elif st in ('S', 'C'):
                    if f.lane in lane_pending:
                        continue
                    if lib.bbx_event_query(f.evS if st == 'S' else f.evC) != 1:
                        lane_pending.add(f.lane)
                        continue
                if st == 'A':
                    f.tA = time.perf_counter()
                    self._submit_fits(f)
                    progressed = True
                elif st == 'B' and f.res.ready():
                    f.tB = time.perf_counter()
                    try:
                        results = f.res.get()
                    except Exception as e:                         # os_corr failed on the host: zero overscan
                        if self.log is not None:
                            self.log.error('frame %d: overscan fits failed (%s); adopting an overscan of zero', f.idx, e)
                        results = None
                    results = self._partial_overscan(f, results)
                    f.state = 'Q'                                  # queued at its lane
                    second = self.two_phase and results is not None and f.os_fail is None
                    self.lane_thread[f.lane].q.put((self._satcol if second else self._device_stage, f, results))
                    progressed = True
                elif st == 'S':
                    self._submit_phase2(f)
                    progressed = True
                elif st == 'B2' and f.res2.ready():
                    try:
                        r2 = f.res2.get()
                        bad = [c for c, r in enumerate(r2) if isinstance(r, dict)]
                        if bad:
                            f.p1[bad[0]] = dict(f.p1[bad[0]], failed=r2[bad[0]]['failed'])
                            f.p1 = self._partial_overscan(f, f.p1)
                    except Exception as e:
                        if self.log is not None:
                            self.log.error('frame %d: overscan fits failed (%s); adopting an overscan of zero', f.idx, e)
                        f.p1 = None
                    f.state = 'Q'
                    self.lane_thread[f.lane].q.put((self._device_stage, f, f.p1))
                    progressed = True
                elif st == 'err':
                    raise f.err
                elif st == 'C':
                    f.tC = time.perf_counter()
                    self.t_stats[0] += f.tA - f.t0
                    self.t_stats[1] += f.tB - f.tA
                    self.t_stats[2] += f.tC - f.tB
                    self.t_stats[3] += 1
                    self._finalize(f)
                    progressed = True
            for f in [x for x in live if x.state == 'done']:
                live.remove(f)
                self.free_slots.append(f.slot)
                ndone += 1
                if on_done:
                    on_done(f.idx, f)
                f.out_group = None                                 # (frame <-> group: no reference cycle left behind)
            if exhausted and not live:
                break
            if not progressed:
                time.sleep(0.0002)
        # device-side error flags were moved into each frame's record (bbx_step_mark): a failing
        # frame is flagged in its own header and does not abort the run
        torch.cuda.synchronize(self.ctx.device)
        return ndone
