#!/usr/bin/env python
"""bench.py -- frames/sec of the per-image reduction hot path on MI355X.

Workload (BASELINE.json configs[1]): one 10600x12000 raw MeerLICHT frame ->
gain/overscan/flat calibration + initial mask (saturation, crosstalk flags,
closing, hole fill) + LA-Cosmic (niter=3), i.e. blackbox_reduce up to and
including cosmics_corr (blackbox.py:1451-1878).  A "step" = one frame.  Inputs
(raw frame, master flat, bad-pixel mask) are synthetic and resident in HBM when
the timed region starts; every step includes the host-side overscan fits and the
two small device<->host hops they need.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--raw u16|f32] [--small]
  N > 1: launched by torch.distributed.run, one rank per GPU; every rank reduces its
  own frames (frames are independent, no data-path collective) -> weak scaling.

Prints ONE JSON line (rank 0) with the roofline of the dominant kernel and a CPU
baseline (the oracle restatement timed on a bounded sub-frame of the same scene).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


def synth_frame_device(torch, dev, ysz, xsz, os_y, os_x, seed, raw_dtype):
    """full-size synthetic raw frame + flat + BPM, generated on the GPU (SURVEY 8d recipe:
    per-channel bias/read noise, sky, stars incl. saturated ones, cosmic-ray tracks)"""
    from blackbox_amd import settings
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    ny, nx = 2 * ysz, 8 * xsz
    gain = torch.tensor(settings.gain['ML1'], dtype=torch.float32, device=dev)
    yy = torch.arange(ny, device=dev, dtype=torch.float32)[:, None]
    xx = torch.arange(nx, device=dev, dtype=torch.float32)[None, :]
    scene = 250.0 + 12.0 * (xx / nx - 0.5) + 8.0 * (yy / ny - 0.5)
    scene = scene.expand(ny, nx).contiguous()
    # stars: Moffat beta=2.5, FWHM 3-5 px, power-law fluxes, ~20 saturating
    nstar = max(20, int(2.0e4 * (ny * nx) / 111513600.0))
    rs = np.random.RandomState(seed)
    sx = rs.uniform(0, nx, nstar); sy = rs.uniform(0, ny, nstar)
    flux = 1e3 * (1.0 - rs.uniform(0, 1, nstar) * (1 - 1e-4)) ** (-1.0)      # 1e3 .. 1e7
    nsat = max(2, nstar // 1000)
    flux[:nsat] = rs.uniform(2e7, 6e7, nsat)
    fwhm = rs.uniform(3.0, 5.0, nstar)
    R = 12
    oy, ox = np.mgrid[-R:R + 1, -R:R + 1]
    for s0 in range(0, nstar, 4096):
        sl = slice(s0, min(nstar, s0 + 4096))
        cx = torch.tensor(sx[sl], device=dev, dtype=torch.float32)[:, None]
        cy = torch.tensor(sy[sl], device=dev, dtype=torch.float32)[:, None]
        fl = torch.tensor(flux[sl], device=dev, dtype=torch.float32)[:, None]
        a = torch.tensor(fwhm[sl] / (2 * np.sqrt(2 ** (1 / 2.5) - 1)), device=dev, dtype=torch.float32)[:, None]
        px = (cx.floor() + torch.tensor(ox.ravel(), device=dev, dtype=torch.float32)[None, :])
        py = (cy.floor() + torch.tensor(oy.ravel(), device=dev, dtype=torch.float32)[None, :])
        r2 = (px - cx) ** 2 + (py - cy) ** 2
        val = fl * 1.5 / (np.pi * a * a) * (1 + r2 / (a * a)) ** (-2.5)
        ok = (px >= 0) & (px < nx) & (py >= 0) & (py < ny)
        idx = (py.long() * nx + px.long())[ok]
        scene.view(-1).index_add_(0, idx, val[ok])
    # cosmic rays: 600 straight tracks (10/s x 60 s), 1 px wide, length 1-30
    ncr = max(10, int(600 * (ny * nx) / 111513600.0))
    cr_idx, cr_val = [], []
    for k in range(ncr):
        x0, y0 = rs.randint(3, nx - 3), rs.randint(3, ny - 3)
        length = rs.randint(1, 31); ang = rs.uniform(0, np.pi)
        amp = rs.uniform(200, 20000)
        for j in range(length):
            x = int(round(x0 + j * np.cos(ang))); y = int(round(y0 + j * np.sin(ang)))
            if 0 <= x < nx and 0 <= y < ny:
                cr_idx.append(y * nx + x); cr_val.append(amp)
    scene.view(-1).index_add_(0, torch.tensor(cr_idx, device=dev), torch.tensor(cr_val, device=dev, dtype=torch.float32))
    scene = scene + torch.sqrt(scene.clamp(min=0)) * torch.randn(ny, nx, device=dev, generator=g)
    # master flat: vignetting + pixel noise
    r2 = ((xx - 0.5 * nx) / (0.5 * nx)) ** 2 + ((yy - 0.5 * ny) / (0.5 * ny)) ** 2
    flat = (1.0 - 0.05 * r2 + 0.005 * torch.randn(ny, nx, device=dev, generator=g)).to(torch.float32).contiguous()
    scene = scene * flat
    bpm = torch.zeros((ny, nx), dtype=torch.uint8, device=dev)
    bpm[torch.rand(ny, nx, device=dev, generator=g) < 2e-4] = 1
    e = 30 if ny > 1000 else 4
    bpm[:e, :] = 32; bpm[-e:, :] = 32; bpm[:, :e] = 32; bpm[:, -e:] = 32
    # assemble the raw frame channel by channel
    dy, dx = ysz + os_y, xsz + os_x
    raw = torch.empty((2 * dy, 8 * dx), dtype=torch.float32, device=dev)
    for c in range(16):
        iy, ix = c // 8, c % 8
        level = 3000.0 + 150.0 * (rs.uniform() - 0.5)
        t = torch.arange(dy, device=dev, dtype=torch.float32)[:, None] / dy - 0.5
        xc = torch.arange(dx, device=dev, dtype=torch.float32)[None, :]
        chan = level + 2.0 * t + 3.0 * t ** 3 + 4.0 * torch.randn(dy, dx, device=dev, generator=g)
        col = 20.0 * torch.exp(-xc / 30.0)
        col[:, xsz:] = 0
        chan = chan + col
        ys0 = 0 if iy == 0 else os_y
        chan[ys0:ys0 + ysz, :xsz] += scene[iy * ysz:(iy + 1) * ysz, ix * xsz:(ix + 1) * xsz] / gain[c]
        raw[iy * dy:(iy + 1) * dy, ix * dx:(ix + 1) * dx] = chan
    raw = raw.round().clamp(0, 65535)
    del scene
    if raw_dtype == 'u16':
        raw = raw.to(torch.int32).to(torch.uint16)
    return raw.contiguous(), flat, bpm


def cpu_baseline(seconds_budget=30.0):
    """oracle (numpy/scipy restatement, single core) on a bounded sub-frame of the same
    kind of scene: 2x8 channels of 660x330 px (1/32 of the frame), scaled by area"""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import bbx_oracle as O
    import lacosmic as L
    from blackbox_amd import settings, synth
    ys, xs = 660, 330
    case = synth.make_case(ys, xs, 4242, tel='ML1', os_y=20, os_x=45, n_stars=200, n_sat=2, n_cr=6)
    t0 = time.perf_counter()
    data = case['raw'].astype('float32')
    gain, sat = settings.gain['ML1'], settings.satlevel['ML1']
    O.gain_corr(data, gain, ys, xs)
    out, h, _ = O.os_corr(data, ys, xs)
    mask, hm = O.mask_init(out, h, case['bpm'], gain, sat, ys, xs)
    out /= case['flat']
    L.detect_cosmics(out, mask != 0, 15, 0.01, 3, 3, h['RDNOISE'])
    dt = time.perf_counter() - t0
    frac = (2 * ys * 8 * xs) / 111513600.0
    return dict(value=frac / dt, unit='frames/s', cores=1, kind='port',
                sample='oracle (numpy/scipy) on a 1320x2640 px sub-frame (1/%.0f of a frame) in %.1f s, scaled by area'
                       % (1 / frac, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=240)
    ap.add_argument('--warmup', type=int, default=12)
    ap.add_argument('--raw', default='u16', choices=['u16', 'f32'])
    ap.add_argument('--small', action='store_true', help='reduced geometry (debug)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--depth', type=int, default=18, help='frames in flight')
    ap.add_argument('--workers', type=int, default=None, help='host fit worker processes')
    ap.add_argument('--lanes', type=int, default=6, help='stage-C lanes (context + stream + issuing thread) per GPU')
    args = ap.parse_args()

    import torch
    from blackbox_amd import reduce as R
    from blackbox_amd import settings

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    # rehearsal knobs (a 1-GPU box): BBX_BENCH_BACKEND=gloo BBX_BENCH_ONE_GPU=1 run all ranks on cuda:0
    backend = os.environ.get('BBX_BENCH_BACKEND', 'nccl')
    if os.environ.get('BBX_BENCH_ONE_GPU'):
        local = 0
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    ctx = R.Context(local)
    dev = ctx.device
    if args.small:
        ysz, xsz, os_y, os_x = 660, 330, 20, 45
    else:
        ysz, xsz, os_y, os_x = 5280, 1320, 20, 180
    raw, flat, bpm = synth_frame_device(torch, dev, ysz, xsz, os_y, os_x, 1000 * 2 + rank, args.raw)
    geom = R.geometry(raw.shape, ysz, xsz)
    tel = 'ML1'
    N = 2 * ysz * 8 * xsz
    nraw = raw.numel()

    import ctypes as C
    from blackbox_amd import _lib
    from blackbox_amd.pipeline import FramePipeline, HostPool

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- serial reference run of one frame (stage breakdown, not the timed region) -----------
    stage_ms = {}

    def frame_serial():
        header, hm = {}, {}
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        R.gain_corr(header, tel)
        sol = R.os_solve(ctx, raw, header, tel, geom)
        ev[1].record()
        data, mask = R.calibrate(ctx, raw, sol, header, hm, tel, geom, mflat=flat, bpm=bpm)
        ev[2].record()
        d_nobj = R.mask_init_finish(ctx, mask, header, hm, geom)
        ev[3].record()
        st = R.cosmics_corr(ctx, data, header, mask, hm, tel)
        ev[4].record()
        ctx.sync()
        return ev, st

    frame_serial()
    # kernels alone on the GPU (no second lane, no other frame in flight): isolated timings of
    # the two dense kernels, reported next to the live ones of the timed region
    _lib.check(_lib.lib.bbx_profile_enable(ctx.h, 1), 'bbx_profile_enable')
    for _ in range(10):
        frame_serial()
    iso_ms = (C.c_double * 8)()
    iso_calls = (C.c_int32 * 8)()
    _lib.check(_lib.lib.bbx_profile_read(ctx.h, iso_ms, iso_calls, 8), 'bbx_profile_read', ctx.h)
    _lib.check(_lib.lib.bbx_profile_enable(ctx.h, 0), 'bbx_profile_enable')
    t0 = time.perf_counter()
    ev, st = frame_serial()
    latency_ms = 1e3 * (time.perf_counter() - t0)
    for i, n in enumerate(['overscan(stats+host fits)', 'calibrate', 'mask_finish', 'lacosmic']):
        stage_ms[n] = ev[i].elapsed_time(ev[i + 1])
    stats = st.cpu().numpy().tolist()

    # ---- timed region: K frames through the pipelined path -------------------------------------
    pool = HostPool(args.workers)
    pipe = FramePipeline(ctx, tel, geom, mflat=flat, bpm=bpm, pool=pool, depth=args.depth, lanes=args.lanes)
    pipe.run([(raw, {}) for _ in range(max(args.warmup, 1))])
    pipe.t_stats = [0.0, 0.0, 0.0, 0]
    check = _lib.check
    check(_lib.lib.bbx_profile_enable(ctx.h, 1), 'bbx_profile_enable')
    barrier()
    t0 = time.perf_counter()
    pipe.run([(raw, {}) for _ in range(args.steps)])
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    nsl = 8
    ms_tot = (C.c_double * nsl)()
    calls = (C.c_int32 * nsl)()
    check(_lib.lib.bbx_profile_read(ctx.h, ms_tot, calls, nsl), 'bbx_profile_read', ctx.h)
    check(_lib.lib.bbx_profile_enable(ctx.h, 0), 'bbx_profile_enable')
    pipe.close()
    pool.close()
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        b_raw = 2 if args.raw == 'u16' else 4
        # algorithmic bytes per launch: calibration reads the raw data sections, flat,
        # BPM and writes data + mask; the dense LA-Cosmic pass reads the frame and the mask once
        # (iterations 2..n work on the surroundings of the cleaned pixels only)
        kern = {
            'k_calibrate': (0, b_raw * N + 4 * N + N + 4 * N + N),
            # the one dense LA-Cosmic pass reads the frame (the mask plane too only while the
            # background-level feed is on, i.e. after a frame needed the level: not in this workload)
            'k_lac_cand': (1, 4 * N),
        }
        # Kernel durations: HIP events recorded by the library around each launch, on the launch
        # stream (lane 0's context carries the timers, so one frame in [lanes] is sampled).  The
        # roofline uses the event pairs of the timed region: the kernel shares the GPU with the
        # other lanes' kernels there, and the rocprofv3 --kernel-trace average of this command
        # shows the same duration.  `isolated` gives the same kernels from the serial frames run
        # in this process just before the timed region (one stream, nothing else in flight):
        # that is the kernel's own efficiency, the timed-region figure its share of a busy GPU.
        iso = {k: (iso_ms[sl] / max(1, iso_calls[sl]), by, iso_calls[sl]) for k, (sl, by) in kern.items()}
        live = {k: (ms_tot[sl] / max(1, calls[sl]), by, calls[sl]) for k, (sl, by) in kern.items()}
        dom = max(live, key=lambda k: live[k][0])          # both run once per frame
        avg_ms, by, ncall = live[dom]

        def gbs(t):
            return t[1] / (t[0] * 1e-3) / 1e9
        roof = dict(bound='hbm', kernel=dom, achieved=gbs(live[dom]), peak=HBM_PEAK_GBS, unit='GB/s',
                    avg_launch_ms=avg_ms, launches=int(ncall), bytes_per_launch=by, traffic=None,
                    others={k: dict(avg_launch_ms=live[k][0], achieved=gbs(live[k]), frac=gbs(live[k]) / HBM_PEAK_GBS)
                            for k in live if k != dom})
        roof['frac'] = roof['achieved'] / roof['peak']
        roof['timing'] = ('HIP events around each launch on the launch stream, timed region, lane 0 of %d '
                          '(other lanes\' kernels run concurrently)' % args.lanes)
        roof['isolated'] = {k: dict(avg_launch_ms=iso[k][0], launches=int(iso[k][2]), achieved=gbs(iso[k]),
                                    frac=gbs(iso[k]) / HBM_PEAK_GBS,
                                    note='serial frames, kernel alone on the GPU') for k in iso}
        # HBM traffic per launch from the committed rocprofv3 --pmc passes of this command
        # (profiles/r01_pmc_traffic.json: FETCH_SIZE/WRITE_SIZE, gfx950 correction applied)
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))['kernels']
            if not args.small and args.raw == 'u16':
                if dom == 'k_calibrate':
                    roof['traffic'] = next(v for k, v in pmc.items() if k.startswith('k_calibrate_v4'))['traffic_bytes_per_launch']
                else:
                    roof['traffic'] = pmc['k_lac_cand_v4<true>']['traffic_bytes_per_launch']
                roof['traffic_source'] = 'profiles/r01_pmc_traffic.json'
        except Exception:
            pass
        out = dict(metric='10560x10560 fp32 frames/sec end-to-end reduce (calibration + LA-Cosmic)',
                   value=args.steps * world / dt, unit='frames/s', n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True,
                   scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload='configs[1]: one %dx%d raw (%s) -> %dx%d frame, gain+overscan+flat+mask_init+LA-Cosmic(niter=3), ML1'
                               % (raw.shape[0], raw.shape[1], args.raw, 2 * ysz, 8 * xsz),
                               frames_per_gpu=args.steps, frames_in_flight=args.depth, stageC_lanes=args.lanes, host_fit_workers=pool.n,
                               parallelism='frame-per-gpu x%d (no collective)' % world),
                   pipeline_wall_ms_per_frame=dict(zip(['stageA_stats', 'stageB_host_fits', 'stageC_device'],
                                                       [1e3 * t / max(1, pipe.t_stats[3]) for t in pipe.t_stats[:3]])),
                   single_frame_latency_ms=latency_ms, stage_ms_serial=stage_ms, lacosmic_stats=stats,
                   device_ms_per_frame_serial={'k_calibrate': iso_ms[0] / max(1, iso_calls[0]),
                                               'k_lac_cand': iso_ms[1] / max(1, iso_calls[1]),
                                               'lac_sparse(x3)': 3 * iso_ms[2] / max(1, iso_calls[2])},
                   roofline=roof)
        if not args.no_cpu:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
