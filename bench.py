#!/usr/bin/env python
"""bench.py -- frames/sec of the per-image reduction hot path on MI355X.

Headline workload (BASELINE.json metric "end-to-end reduce+ZOGY", the per-frame work of
configs[4] on one GPU): one 10600x12000 raw MeerLICHT frame -> gain / overscan / flat
calibration + initial mask + LA-Cosmic (niter=3) + crosstalk + satellite trail + mask counts +
edge fill (blackbox_reduce, blackbox.py:1451-1974), then zogy.optimal_subtraction against a
co-added reference of the field (call site 2460-2465): background mesh + subtraction, variance
images, 64 sub-images of 1400^2 through the ZOGY FFT subtraction, D / Scorr / Fpsf / Fpsferr,
transient candidates, and the full-source catalogue by PSF-weighted optimal photometry.
A "step" = one frame.  Inputs (raw frames, master flat, bad-pixel mask, crosstalk
coefficients, reference image, PSFs) are synthetic and resident in HBM when the timed
region starts; every step includes the host-side overscan fits and the small device<->host
hops of the path.  --workload selects configs[1] (calibration + LA-Cosmic) or configs[2]
(full calibration + background mesh + satellite trail) as the timed workload instead; their
rates are also measured briefly and attached to the headline line (other_workloads).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload zogy|full|calib] [--small]
  N > 1: one rank per GPU (torch.distributed.run; `--gpus N` alone spawns it); every rank
  reduces its own frames (independent, no data-path collective) -> weak scaling.

Timing: depth + W + K + depth frames go through the frames-in-flight pipeline in one run
(depth = frames in flight).  The clock runs from the completion of the last warm-up frame to the
completion of the K-th frame after it: exactly K completions with the pipeline full at both
ends (neither its fill nor its drain inside the region; barrier + device synchronisation
around the run).  `idle_to_idle` is the rate of all those frames from an idle device to an idle
device, `long_run` the same steady-state measurement over 240 frames.

Prints ONE JSON line (rank 0) with the roofline of the dominant kernel and a CPU baseline.

Beside the headline (N = 1, default run): `roofline.stages` (the calibration and full-calibration workloads),
`io_inclusive.measured` = files to files through the operator's own entry -- a child `python blackbox.py --image_list`
over 96 full-size fpacked raws, every product written, once onto the RAM disk and once onto local scratch (after an
untimed 48-file warm-up run) --, `process_per_file` (one `python blackbox.py --image` process), `process_pool`
(`--nproc 4`), `cpu_baseline` (the oracle on the box's cores, checked against one whole frame).
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


NSL = 14                                                         # BBX_PROF_NSLOTS (launch slots of bbx_profile_read)


def moffat_stamp(S, fwhm, beta=2.5):
    """unit-sum Moffat PSF stamp [S, S] (float32), centre at S // 2"""
    a = fwhm / (2 * np.sqrt(2 ** (1 / beta) - 1))
    y, x = np.mgrid[0:S, 0:S] - S // 2
    p = (1 + (y * y + x * x) / (a * a)) ** -beta
    return (p / p.sum()).astype(np.float32)


def synth_frame_device(torch, dev, ysz, xsz, os_y, os_x, seed, raw_dtype, extras=False, ntrans=0, trail=None, psf_field=None):
    """full-size synthetic raw frame + flat + BPM, generated on the GPU (SURVEY 8d recipe:
    per-channel bias/read noise, sky, stars incl. saturated ones, cosmic-ray tracks).
    extras: also return dict(scene0 = the noiseless sky + stars scene in e- (what a deep reference
    image of the field shows), transients = [(y, x, flux)] of [ntrans] point sources (Moffat FWHM 4)
    added to this frame only, S/N 6-100); trail = (xa, ya, xb, yb, amp, fwhm): a satellite trail.
    psf_field = dict(size, nsx, fw_n, fw_r, fratio) (zogy_psf_field): the scene follows the PSF model that ZOGY is handed --
    a star (and a transient) of the new frame has the FWHM of the sub-image it lies in, the reference scene (scene0) shows
    the same stars with the reference's FWHM there and their fluxes divided by the flux ratio, wings out to 24 px.  Without
    it every star draws its own FWHM (3-5 px) in both images: each bright star then leaves a PSF-mismatch residual in the
    subtraction (round 4's bench scene: ~1900 such detections beside its 50 transients)."""
    from blackbox_amd import settings
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    ny, nx = 2 * ysz, 8 * xsz
    gain = torch.tensor(settings.gain['ML1'], dtype=torch.float32, device=dev)
    yy = torch.arange(ny, device=dev, dtype=torch.float32)[:, None]
    xx = torch.arange(nx, device=dev, dtype=torch.float32)[None, :]
    scene = sky_model(torch, dev, ny, nx)
    # stars: Moffat beta=2.5, FWHM 3-5 px, power-law fluxes, ~20 saturating
    nstar = max(20, int(2.0e4 * (ny * nx) / 111513600.0))
    rs = np.random.RandomState(seed)
    sx = rs.uniform(0, nx, nstar); sy = rs.uniform(0, ny, nstar)
    flux = 1e3 * (1.0 - rs.uniform(0, 1, nstar) * (1 - 1e-4)) ** (-1.0)      # 1e3 .. 1e7
    nsat = max(2, nstar // 1000)
    flux[:nsat] = rs.uniform(2e7, 6e7, nsat)
    fwhm = rs.uniform(3.0, 5.0, nstar)
    R = 12 if psf_field is None else 24

    def add_stars(img, fw, fl_all):
        oy, ox = np.mgrid[-R:R + 1, -R:R + 1]
        chunk = 4096 if R <= 12 else 1024
        for s0 in range(0, nstar, chunk):
            sl = slice(s0, min(nstar, s0 + chunk))
            cx = torch.tensor(sx[sl], device=dev, dtype=torch.float32)[:, None]
            cy = torch.tensor(sy[sl], device=dev, dtype=torch.float32)[:, None]
            fl = torch.tensor(fl_all[sl], device=dev, dtype=torch.float32)[:, None]
            a = torch.tensor(fw[sl] / (2 * np.sqrt(2 ** (1 / 2.5) - 1)), device=dev, dtype=torch.float32)[:, None]
            px = (cx.floor() + torch.tensor(ox.ravel(), device=dev, dtype=torch.float32)[None, :])
            py = (cy.floor() + torch.tensor(oy.ravel(), device=dev, dtype=torch.float32)[None, :])
            r2 = (px - cx) ** 2 + (py - cy) ** 2
            val = fl * 1.5 / (np.pi * a * a) * (1 + r2 / (a * a)) ** (-2.5)
            ok = (px >= 0) & (px < nx) & (py >= 0) & (py < ny)
            idx = (py.long() * nx + px.long())[ok]
            img.view(-1).index_add_(0, idx, val[ok])
    scene0 = None
    if psf_field is None:
        add_stars(scene, fwhm, flux)
        scene0 = scene.clone() if extras else None
    else:
        ksub = (np.minimum(sy.astype(int), ny - 1) // psf_field['size']) * psf_field['nsx'] + np.minimum(sx.astype(int), nx - 1) // psf_field['size']
        if extras:
            scene0 = scene.clone()
            add_stars(scene0, np.asarray(psf_field['fw_r'])[ksub], flux / np.asarray(psf_field['fratio'])[ksub])
        add_stars(scene, np.asarray(psf_field['fw_n'])[ksub], flux)
    transients = []
    if ntrans:
        rt = np.random.RandomState(seed + 77)
        S = 25
        st = torch.tensor(moffat_stamp(S, 4.0), device=dev).reshape(-1)
        oyx = np.mgrid[-(S // 2):S // 2 + 1, -(S // 2):S // 2 + 1]
        for _ in range(ntrans):
            ty, tx = int(rt.randint(40, ny - 40)), int(rt.randint(40, nx - 40))
            fl = float(110.0 * 10 ** rt.uniform(np.log10(6), 2))               # S/N 6 .. 100 for ~110 e- of noise per PSF
            idx = torch.tensor(((ty + oyx[0]) * nx + (tx + oyx[1])).ravel(), device=dev)
            if psf_field is not None:
                st = torch.tensor(moffat_stamp(S, float(psf_field['fw_n'][(ty // psf_field['size']) * psf_field['nsx'] + tx // psf_field['size']])),
                                  device=dev).reshape(-1)
            scene.view(-1).index_add_(0, idx, st * fl)
            transients.append((ty, tx, fl))
    if trail is not None:
        xa, ya, xb, yb, amp, width = trail
        norm = float(np.hypot(xb - xa, yb - ya))
        for y0 in range(0, ny, 1024):                                          # in row bands: one distance plane at a time
            yb_ = min(ny, y0 + 1024)
            d = ((xx - xa) * (yb - ya) - (yy[y0:yb_] - ya) * (xb - xa)) / norm
            scene[y0:yb_] += amp * torch.exp(-0.5 * (d / (width / 2.355)) ** 2)
        del d
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
    if extras:
        return raw.contiguous(), flat, bpm, dict(scene0=scene0, transients=transients)
    return raw.contiguous(), flat, bpm


def sky_model(torch, dev, ny, nx):
    """the smooth sky of the synthetic scene [e-]"""
    yy = torch.arange(ny, device=dev, dtype=torch.float32)[:, None]
    xx = torch.arange(nx, device=dev, dtype=torch.float32)[None, :]
    return (250.0 + 12.0 * (xx / nx - 0.5) + 8.0 * (yy / ny - 0.5)).expand(ny, nx).contiguous()


def synth_reference(torch, dev, scene0, seed, depth=4.0, sky_ref=0.0):
    """the co-added reference image of the field: the same scene without
    cosmic rays / trails / transients at [depth] x the exposure (noise / sqrt(depth) in the
    new frame's units); background-subtracted like a buildref product (sky_ref = 0) or with
    a flat sky level of its own; mask all zero"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 4242)
    ref = scene0 - sky_model(torch, dev, *scene0.shape) + sky_ref
    ref = ref + torch.sqrt((scene0 / depth).clamp(min=0)) * torch.randn(scene0.shape, device=dev, generator=g)
    return ref.contiguous(), torch.zeros(scene0.shape, dtype=torch.uint8, device=dev)


def zogy_psf_field(nsy, nsx, size):
    """the PSF model of the synthetic field, per sub-image: FWHM of the new frame (3.6 .. 4.6 px) and of the co-added reference
    (3.3 .. 3.9 px), flux ratio new / reference -- what zogy_inputs turns into stamps and scalars and synth_frame_device into
    stars"""
    nsub = nsy * nsx
    ky, kx = np.divmod(np.arange(nsub), nsx)
    u, v = kx / max(1, nsx - 1) - 0.5, ky / max(1, nsy - 1) - 0.5
    rs = np.random.RandomState(5)
    return dict(size=size, nsx=nsx, fw_n=4.1 + 0.6 * u + 0.37 * v, fw_r=3.6 + 0.3 * u - 0.29 * v,
                fratio=1.0 + 0.04 * u - 0.03 * v + rs.normal(0, 0.005, nsub))


def zogy_inputs(torch, dev, nsy, nsx, S, box, ny, nx):
    """what zogy hands run_ZOGY per sub-image (blackbox.py:3754-3759, call 2460-2465) -- analytic Moffat
    PSF stamps S x S (49 x 49), one pair per sub-image with a FWHM gradient across the field (new 3.6 .. 4.6 px, co-added
    reference 3.3 .. 3.9 px), flux ratio and astrometric scatter per sub-image; the reference is a co-add:
    background-subtracted, with its `_bkg_std_mini` image (buildref products), a smooth 7.6 .. 8.4 e- here"""
    nsub = nsy * nsx
    ky, kx = np.divmod(np.arange(nsub), nsx)
    u, v = kx / max(1, nsx - 1) - 0.5, ky / max(1, nsy - 1) - 0.5
    fw_n, fw_r = 4.1 + 0.6 * u + 0.37 * v, 3.6 + 0.3 * u - 0.29 * v
    psf_n = torch.from_numpy(np.stack([moffat_stamp(S, f) for f in fw_n])).to(dev)
    psf_r = torch.from_numpy(np.stack([moffat_stamp(S, f) for f in fw_r])).to(dev)
    rs = np.random.RandomState(5)
    by, bx = np.mgrid[0:ny // box, 0:nx // box]
    ref_std_mini = (8.0 + 0.4 * (bx / (nx // box) - 0.5) - 0.4 * (by / (ny // box) - 0.5)).astype(np.float32)
    return dict(psf_new=psf_n, psf_ref=psf_r, fratio=1.0 + 0.04 * u - 0.03 * v + rs.normal(0, 0.005, nsub),
                dx=0.03 + 0.01 * rs.uniform(-1, 1, nsub), dy=0.03 + 0.01 * rs.uniform(-1, 1, nsub), ref_is_bkgsub=True,
                ref_bkg_std_mini=ref_std_mini)


CPU_SAMPLE = (660, 330)                 # channel size of the CPU sample: 16 channels = a 1320 x 2640 sub-frame, 1/32 of a frame


def _cpu_sample(args):
    """one worker's bounded sample of the CPU path: the oracle restatement of the reduction (numpy / scipy; LA-Cosmic by the
    oracle's C twin oracle/lacosmic_c.c -- compiled C like the astroscrappy the reference calls, blackbox.py:4323-4332 -- on
    [nthreads] OpenMP threads) on a 1320x2640 sub-frame (1/32 of a frame), the background mesh on it, and run_zogy
    (numpy FFTs) on one 1400^2 sub-image; returns the seconds of each part"""
    seed, with_zogy, nthreads = args
    for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ[k] = str(nthreads)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import bbx_oracle as O
    import lacosmic_c as LC
    import zogy_core as Z
    from blackbox_amd import settings, synth
    ys, xs = CPU_SAMPLE
    case = synth.make_case(ys, xs, 4242 + seed, tel='ML1', os_y=20, os_x=45, n_stars=200, n_sat=2, n_cr=6)
    t0 = time.perf_counter()
    data = case['raw'].astype('float32')
    gain, sat = settings.gain['ML1'], settings.satlevel['ML1']
    O.gain_corr(data, gain, ys, xs)
    out, h, _ = O.os_corr(data, ys, xs)
    mask, hm = O.mask_init(out, h, case['bpm'], gain, sat, ys, xs)
    out /= case['flat']
    LC.detect_cosmics(out, mask != 0, 15, 0.01, 3, 3, h['RDNOISE'], nthreads=nthreads)
    t1 = time.perf_counter()
    t_bkg = t_zogy = 0.0
    if with_zogy:
        med, std = Z.get_back_mini(out, mask, None, box=60)
        med, std = Z.fill_filter_mini(med), Z.fill_filter_mini(std)
        Z.mini2back(med, out.shape, 60)
        Z.mini2back(std, out.shape, 60)
        t2 = time.perf_counter()
        rs = np.random.RandomState(seed)
        Ls = 1400
        img = rs.normal(0, 20, (Ls, Ls)).astype('float32')
        psf = np.zeros((Ls, Ls), 'float32'); st = moffat_stamp(25, 4.0)
        for j in range(25):
            for i in range(25):
                psf[(j - 12) % Ls, (i - 12) % Ls] = st[j, i]
        V = np.full((Ls, Ls), 400.0, 'float32')
        Z.run_zogy(img, img[::-1].copy(), psf, psf, 20.0, 10.0, 1.0, 1.0, V, V, 0.03, 0.03)
        t3 = time.perf_counter()
        t_bkg, t_zogy = t2 - t1, t3 - t2
    return (t1 - t0, t_bkg, t_zogy)


def _cpu_full_frame(args):
    """the same parts as _cpu_sample on ONE WHOLE frame of the bench's own workload (the raw frame, flat and bad-pixel mask the
    GPU run used, handed over as .npy files), one process, one thread: what the area extrapolation of the sample is checked
    against.  -> seconds of (reduction, mesh of the new frame, run_zogy on the 64 sub-images)"""
    path, with_zogy = args
    for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ[k] = '1'
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import bbx_oracle as O
    import lacosmic_c as LC
    import zogy_core as Z
    from blackbox_amd import settings
    raw, flat, bpm = (np.load(path + k + '.npy') for k in ('raw', 'flat', 'bpm'))
    ys, xs = 5280, 1320
    t0 = time.perf_counter()
    data = raw.astype('float32')
    gain, sat = settings.gain['ML1'], settings.satlevel['ML1']
    O.gain_corr(data, gain, ys, xs)
    out, h, _ = O.os_corr(data, ys, xs)
    mask, hm = O.mask_init(out, h, bpm, gain, sat, ys, xs)
    out /= flat
    LC.detect_cosmics(out, mask != 0, 15, 0.01, 3, 3, h['RDNOISE'], nthreads=1)
    t1 = time.perf_counter()
    t_bkg = t_zogy = 0.0
    if with_zogy:
        med, std = Z.get_back_mini(out, mask, None, box=60)
        med, std = Z.fill_filter_mini(med), Z.fill_filter_mini(std)
        sub = (out - Z.mini2back(med, out.shape, 60)).astype('float32')
        sig = Z.mini2back(std, out.shape, 60)
        t2 = time.perf_counter()
        Ls = 1400
        psf = np.zeros((Ls, Ls), 'float32'); st = moffat_stamp(25, 4.0)
        for j in range(25):
            for i in range(25):
                psf[(j - 12) % Ls, (i - 12) % Ls] = st[j, i]
        N_s, V_s = Z.cut_subimages(sub, 1320, 40), Z.cut_subimages((np.maximum(sub, 0) + sig * sig).astype('float32'), 1320, 40)
        for k in range(N_s.shape[0]):
            Z.run_zogy(N_s[k], N_s[k][::-1].copy(), psf, psf, 20.0, 10.0, 1.0, 1.0, V_s[k], V_s[k], 0.03, 0.03)
        t3 = time.perf_counter()
        t_bkg, t_zogy = t2 - t1, t3 - t2
    return (t1 - t0, t_bkg, t_zogy)


def cpu_baseline(workload, full_frame=None):
    """the CPU restatement (oracle: numpy / scipy, kind "port") timed on the host cores of this
    box the way the reference farms frames (blackbox.py:363-379: one process per frame, each
    single-threaded): N_proc = min(cores, RAM / 6 GB) workers run the same bounded sample
    concurrently; frames/s = N_proc / (per-frame seconds extrapolated from the sample by area)."""
    import multiprocessing as mp
    from blackbox_amd.pipeline import cpu_budget
    cores = cpu_budget()
    try:
        ram_gb = os.sysconf('SC_PAGE_SIZE') * os.sysconf('SC_PHYS_PAGES') / 2 ** 30
    except (ValueError, OSError):
        ram_gb = 64.0
    nproc = int(max(1, min(cores, ram_gb // 6)))
    with_zogy = workload == 'zogy'
    frac = (2 * CPU_SAMPLE[0] * 8 * CPU_SAMPLE[1]) / 111513600.0

    def frame_seconds(t):
        # reduction and mesh scale with the area; ZOGY with the 64 sub-images; the mesh runs on the new frame only (the
        # headline's reference is a background-subtracted co-add with its bkg_std_mini, as buildref delivers it)
        return t[0] / frac + (t[1] / frac if workload != 'calib' else 0.0) + (64 * t[2] if with_zogy else 0.0)
    # mode (ii) of SURVEY 8d: one process, all cores (the threads go to the C / OpenMP LA-Cosmic, as in the reference's
    # environment; numpy's element-wise stages and FFTs stay on one core) -- and the same sample on one thread
    with mp.get_context('spawn').Pool(1) as pool:
        t1 = pool.map(_cpu_sample, [(0, with_zogy, 1)])[0]
        t_all = pool.map(_cpu_sample, [(0, with_zogy, cores)])[0]
    # mode (i): N_proc single-threaded processes side by side (how the reference farms frames, blackbox.py:363-379)
    with mp.get_context('spawn').Pool(nproc) as pool:
        ts = pool.map(_cpu_sample, [(k, with_zogy, 1) for k in range(nproc)])
    per_frame_all = float(np.mean([frame_seconds(t) for t in ts]))
    full = None
    if full_frame is not None:
        # the extrapolation checked once against a whole frame (one process, one thread, nothing else running)
        try:
            with mp.get_context('spawn').Pool(1) as pool:
                tf = pool.map(_cpu_full_frame, [(full_frame, with_zogy)])[0]
            sec_full, sec_extra = float(sum(tf)), float(frame_seconds(t1))
            full = dict(seconds=round(sec_full, 2), seconds_per_part=[round(float(v), 2) for v in tf],
                        extrapolated_seconds=round(sec_extra, 2), extrapolated_seconds_per_part=[
                            round(float(t1[0] / frac), 2), round(float(t1[1] / frac), 2), round(float(64 * t1[2]), 2)],
                        ratio_full_over_extrapolated=round(sec_full / sec_extra, 3),
                        note='one single-threaded process on the whole 10600x12000 raw frame of the GPU run (reduction, mesh, run_zogy on '
                             'its 64 sub-images) against the same process on the 1/32-frame sample scaled by area (+ one sub-image x 64)')
        except Exception as e:                                       # the validation is an extra: its failure is reported, not fatal
            full = dict(error=repr(e))
    return dict(value=nproc / per_frame_all, full_frame_check=full, unit='frames/s', cores=nproc, kind='port',
                one_core_frames_per_s=1.0 / frame_seconds(t1),
                one_process_all_cores=dict(frames_per_s=1.0 / frame_seconds(t_all), threads=cores,
                                           seconds_per_part=[round(float(v), 2) for v in t_all],
                                           note='SURVEY 8d mode (ii): one process, OMP_NUM_THREADS = cores (LA-Cosmic in C / OpenMP uses them; '
                                                'the numpy stages and FFTs run on one core)'),
                sample='oracle per worker: reduction (numpy / scipy; LA-Cosmic = oracle/lacosmic_c.c, vectorised sorting-network '
                       'medians)%s on a 1320x2640 px sub-frame (1/%.0f of a frame, scaled by '
                       'area)%s; %d single-threaded worker processes at once (cpu budget %d cores, %.0f GB RAM); mean '
                       'seconds per part %s' % (' + background mesh' if workload != 'calib' else '', 1 / frac,
                                                ' + run_zogy on one 1400^2 sub-image (x64)' if with_zogy else '', nproc, cores,
                                                ram_gb, [round(float(np.mean([t[i] for t in ts])), 2) for i in range(3)]),
                reference_own_code_anchor='SURVEY.md section 6 / BASELINE.md section 2: the reference\'s own gain+os_corr+flat+'
                                          'mask_init+xtalk+mask_header+edge fill take ~50 s per frame on one process '
                                          '(no LA-Cosmic, no ZOGY: packages absent)')


def run_pipeline(torch, ctx, tel, geom, raws, kw, steps, warmup, depth, lanes, pool, barrier, prof_ctx=None, outdir=None, nwriters=8,
                 raw_files=None, nreaders=3):
    """Steady-state rate of a FramePipeline: depth + W + K + depth frames go through it in one run, fed
    continuously.  The first [depth] frames fill the pipeline, then W warm-up frames; the clock runs from the
    completion of the last warm-up frame to the completion of the K-th frame after it -- exactly K completions with
    the pipeline full at both ends ([depth] cool-down frames are still in flight behind the last timed one, so the
    timed region holds neither the fill nor the drain; a region that ends with the drain reads too fast, because
    the frames in flight at its start were partly done already).  barrier + device synchronisation before the run
    and after it.
    outdir: run the output stage as well (outstage.OutputStage: every image product of a frame tile-compressed on its
    lane and written as .fits.fz by writer threads, the small products by the frame's callback); a frame then counts
    as complete when its last file is on disk (files are removed again at once: the directory may be a RAM disk).
    raw_files: the input side as well (instage.InputStage): the timed frames are read from these `.fits(.fz)` files (cycled) by
    reader threads, decoded on the device (bbx_funpack_tiles) and handed to the pipeline with their ready events -- the
    reference's read_hdulist (blackbox.py:1451) inside the measured loop.
    -> dict(dt, dt_all (idle to idle, all frames), ...)"""
    from blackbox_amd import _lib
    from blackbox_amd.pipeline import FramePipeline
    mark = {'n': 0, 'bytes': 0, 'files': 0, 'err': None}
    first = depth + warmup                      # completions before the clock starts
    n_all = first + steps + depth
    import threading
    lock, all_done = threading.Lock(), threading.Event()

    def count(f):
        with lock:
            mark['n'] += 1
            n = mark['n']
            if n == first:
                mark['t0'] = time.perf_counter()
                for h in mark.get('prof_h', ()):
                    _lib.check(_lib.lib.bbx_profile_enable(h, 1), 'bbx_profile_enable')
            elif n == first + steps:
                mark['t1'] = time.perf_counter()
                for h in mark.get('prof_h', ()):
                    _lib.check(_lib.lib.bbx_profile_enable(h, 2), 'bbx_profile_enable')      # pause, keep the records
            mark['last'] = f
            if n == n_all:
                all_done.set()
    stage = None
    extra = {}
    if outdir is not None:
        from blackbox_amd import fitsio, outstage, zogy as G
        ny, nx = 2 * geom.ysize_chan, 8 * geom.xsize_chan
        stage = outstage.OutputStage(ctx.device, ny, nx, nwriters=nwriters)

        def on_written(f, group):
            """a writer thread, after the last image of the frame: the small products (mini images, tables, header files),
            then the frame counts; its files are removed again"""
            t_w, t_c = time.perf_counter(), time.thread_time()
            try:
                if group.error is not None:
                    raise group.error
                base = f.out_base
                small = []
                if f.sub is not None:
                    hb = {'BKG-SIZE': f.sub['header_new']['BKG-SIZE']}
                    for name in ('bkg_mini_new', 'bkg_std_mini_new'):
                        p = base + '_' + name.replace('_new', '') + '.fits'
                        fitsio.write_image(p, f.sub[name], hb); small.append(p)
                    if f.sub.get('catalog') is not None:
                        small.append(G.format_cat(f.sub['catalog'], base + '_cat.fits', cat_type='new', header2add=f.header))
                    if f.sub.get('transients') is not None and 'D' in f.out_names:
                        small.append(G.format_cat(G.transient_table(f.sub['transients']), base + '_trans.fits', cat_type='trans',
                                                  header2add=f.header))
                fitsio.write_header(base + '_hdr.fits', f.header); small.append(base + '_hdr.fits')
                nb = 0
                for p in list(group.paths) + [q for q in small if q]:
                    nb += os.path.getsize(p)
                    os.unlink(p)
                with lock:
                    mark['bytes'] += nb
                    mark['files'] += len(group.paths) + len(small)
                    mark['small_wall'] = mark.get('small_wall', 0.0) + time.perf_counter() - t_w
                    mark['small_cpu'] = mark.get('small_cpu', 0.0) + time.thread_time() - t_c
            except BaseException as e:
                mark['err'] = e
            count(f)
        extra = dict(outstage=stage, out_base=lambda idx, h: os.path.join(outdir, 'ML1_f%06d_red' % idx), on_written=on_written)
    pipe = FramePipeline(ctx, tel, geom, pool=pool, depth=depth, lanes=lanes, **dict(kw, **extra))
    if prof_ctx is not None:
        mark['prof_h'] = [lc.h for lc in pipe.lane_ctx]        # the launches of every lane are timed (prof_ctx = lane 0's context)
    # untimed: first-use allocations, rocFFT plans, workspace growth of every lane
    save = (mark['n'], first)
    first = 10 ** 9                             # (the warm-up frames of the pipeline do not count)
    pipe.run([(raws[i % len(raws)], {}) for i in range(max(lanes, 2))])
    if stage is not None:
        t_end = time.time() + 120
        while mark['n'] < max(lanes, 2) and time.time() < t_end:
            time.sleep(0.01)
    mark['n'], first = 0, save[1]
    mark['bytes'], mark['files'] = 0, 0
    pipe.t_stats = [0.0, 0.0, 0.0, 0]
    pipe.lane_cpu = [0.0, 0.0, 0]
    t_cpu_main = time.thread_time()

    src = None
    if raw_files is not None:
        from blackbox_amd import instage
        src = instage.InputStage(ctx, lambda i: raw_files[i % len(raw_files)] if i < n_all else None, (geom.ny_raw, geom.nx_raw),
                                 nreaders=nreaders, nbuf=depth + 4, ahead=4)

    def on_done(idx, f):
        if src is not None:
            src.release(f.raw)                   # the device is through with the frame's raw buffer
        if stage is None:
            count(f)
    barrier()
    wait0 = list(_lib.WAIT_STATS)
    t_all0 = time.perf_counter()
    pipe.run(src if src is not None else [(raws[i % len(raws)], {}) for i in range(n_all)], on_done=on_done)
    torch.cuda.synchronize()
    if stage is not None and not all_done.wait(300.0):
        raise RuntimeError('output stage: %d of %d frames written (%r)' % (mark['n'], n_all, mark['err']))
    barrier()
    t_end = time.perf_counter()
    if mark['err'] is not None:
        raise mark['err']
    out = dict(dt=mark['t1'] - mark['t0'], dt_all=t_end - t_all0, n_all=n_all, t_stats=list(pipe.t_stats), last=mark.get('last'),
               nworkers=pipe.pool.n, bytes_written=mark['bytes'], files_written=mark['files'],
               lane_fetch_wait_ms_per_frame=1e3 * (_lib.WAIT_STATS[0] - wait0[0]) / max(1, n_all), lane_fetch_waits_per_frame=(_lib.WAIT_STATS[1] - wait0[1]) / max(1, n_all),
               host_ms_per_frame=dict(lane_threads_cpu=1e3 * pipe.lane_cpu[0] / max(1, pipe.lane_cpu[2]),
                                      lane_threads_wall=1e3 * pipe.lane_cpu[1] / max(1, pipe.lane_cpu[2]),
                                      orchestrator_cpu=1e3 * (time.thread_time() - t_cpu_main) / n_all))
    if prof_ctx is not None:
        # per-launch-slot totals over the timed region, all lanes
        import ctypes as C
        ms_sum, n_sum = [0.0] * NSL, [0] * NSL
        for h in mark['prof_h']:
            ms, nc = (C.c_double * NSL)(), (C.c_int32 * NSL)()
            _lib.check(_lib.lib.bbx_profile_read(h, ms, nc, NSL), 'bbx_profile_read', h)
            _lib.check(_lib.lib.bbx_profile_enable(h, 0), 'bbx_profile_enable')
            for k in range(NSL):
                ms_sum[k] += ms[k]; n_sum[k] += nc[k]
        out['prof'] = (ms_sum, n_sum)
    pipe.close()
    if stage is not None:
        stage.close()
        out['writer_ms_per_image'] = {k: dict(wall=1e3 * w / max(1, n), cpu=1e3 * c / max(1, n)) for k, (w, c, n) in stage.phase.items()}
        out['lane_slot_wait_ms_per_frame'] = 1e3 * sum(l.slot_wait for l in stage.lanes.values()) / max(1, n_all)
        out['small_files_ms_per_frame'] = dict(wall=1e3 * mark.get('small_wall', 0.0) / max(1, n_all), cpu=1e3 * mark.get('small_cpu', 0.0) / max(1, n_all))
    if src is not None:
        out['bytes_read'] = src.bytes_read
        src.close()
    return out


def measure_io(torch, ctx, tel, geom, raws, kw, depth, lanes, pool, barrier, args, section=lambda name: None, frames=40):
    """SURVEY 8d timing item (iii), measured: the pipeline from raw files to product files (RAM disk and local scratch)"""
    # ---- measured: the same pipeline with the output stage in the loop (SURVEY 8d timing item iii) ------------
    meas = {}
    import shutil
    import tempfile
    simple = getattr(args, 'io_simple', False)                    # (profiling: one run, RAM disk only)
    for label, root in (('ramdisk', '/dev/shm'), ('scratch', tempfile.gettempdir())):
        if not os.path.isdir(root) or not os.access(root, os.W_OK) or (simple and label != 'ramdisk'):
            continue
        td = tempfile.mkdtemp(prefix='bbx_bench_out_', dir=root)
        try:
            # the raw frames as the telescope delivers them: fpacked uint16 (`.fits.fz`, lossless Rice), one file per
            # distinct raw buffer, read back in turn
            from blackbox_amd import fpack as P
            raw_files = []
            for i, rw in enumerate(raws[:8]):
                raw_files.append(P.fpack_image(ctx, os.path.join(td, 'raw_%02d.fits' % i), rw, {'EXPTIME': 60.0, 'OBJECT': 'synthetic'}))
            raw_mb = sum(os.path.getsize(p_) for p_ in raw_files) / len(raw_files) / 1e6
            r4 = run_pipeline(torch, ctx, tel, geom, raws, kw, frames, 4, depth, lanes, pool, barrier, outdir=td, nwriters=args.writers,
                              raw_files=raw_files, nreaders=args.readers)
            meas[label] = dict(frames_per_s=frames / r4['dt'], ms_per_frame=1e3 * r4['dt'] / frames, dir=root,
                               MB_per_frame=r4['bytes_written'] / max(1, r4['n_all']) / 1e6,
                               files_per_frame=r4['files_written'] / max(1, r4['n_all']), writer_threads=args.writers,
                               reader_threads=args.readers, raw_MB_per_frame=raw_mb,
                               host_ms_per_frame=r4['host_ms_per_frame'], writer_ms_per_image=r4.get('writer_ms_per_image'),
                               lane_slot_wait_ms_per_frame=r4.get('lane_slot_wait_ms_per_frame'),
                               lane_fetch_wait_ms_per_frame=r4.get('lane_fetch_wait_ms_per_frame'), lane_fetch_waits_per_frame=r4.get('lane_fetch_waits_per_frame'),
                               small_files_ms_per_frame=r4.get('small_files_ms_per_frame'))
            if not simple:
                r5 = run_pipeline(torch, ctx, tel, geom, raws, kw, frames, 4, depth, lanes, pool, barrier, outdir=td, nwriters=args.writers)
                meas[label]['output_side_only'] = dict(frames_per_s=frames / r5['dt'], note='inputs resident in HBM (round 3 figure)')
        except Exception as e:
            import traceback
            meas[label] = dict(error=repr(e), trace=traceback.format_exc()[-1500:])
        finally:
            shutil.rmtree(td, ignore_errors=True)
        section('io_inclusive.measured ' + label)
    meas['note'] = ('the timed pipeline from files to files: every raw frame read from its fpacked `.fits.fz` file by reader '
                    'threads (pinned buffer, one H2D copy, Rice decode on the device: bbx_funpack_tiles), every product of the frame '
                    'on disk before the frame counts: _red, _mask, _D, _Scorr, '
                    '_Fpsf, _trans_limmag tile-compressed on the lane that made them (bbx_fpack_body, q = 16 / lossless / 16 / 2 / '
                    '4 / 2) and written as .fits.fz by writer threads; _bkg_mini, _bkg_std_mini, _cat, _trans, _hdr by the '
                    'frame\'s callback; steady state over 40 frames; product files removed as soon as written')
    return meas


def cli_image_list(ctx, td, cmd1, raws, hdr, nfiles):
    """files to files through the operator's own entry: a child `python blackbox.py --image_list L ... --fpack True` over [nfiles]
    full-size fpacked raw frames on the RAM disk (blackbox_slurm_google.py:305-309 is the contract), every product of every
    frame written.  The child reports when each file's products were all on disk (BBX_TIMING files_done_unix): the
    steady-state rate is taken between the completions of file [skip] and the last one (the pipeline is full by then, at
    most 16 frames in flight), next to the rate of the whole list from the first input to the last product and the
    process's wall time (imports, GPU context, masters into HBM included)."""
    import shutil
    import subprocess
    from blackbox_amd import fpack as P
    free = shutil.disk_usage(td).free
    need = nfiles * (100e6 + 400e6) + 2e9
    if free < need:
        return dict(skipped='%.1f GB free in %s, %.1f GB needed' % (free / 1e9, td, need / 1e9))
    files = []
    for k in range(nfiles):
        h = dict(hdr, **{'DATE-OBS': '2024-01-02T03:%02d:%02d' % (k // 60, k % 60)})
        files.append(P.fpack_image(ctx, os.path.join(td, 'ML1_list_%03d.fits' % k), raws[k % len(raws)], h))
    ctx.sync()
    lst = os.path.join(td, 'list.txt')
    with open(lst, 'w') as f:
        f.write('\n'.join(files) + '\n')
    import torch
    import gc
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                                      # (the parent's cached blocks: the children need the HBM)
    free_b, total_b = torch.cuda.mem_get_info()
    cmd = [c for c in cmd1]
    i = cmd.index('--image')
    cmd[i:i + 2] = ['--image_list', lst]
    prof = os.environ.get('BBX_CLI_PROFILE')                      # (debug: cProfile of the child's orchestrating thread -> stderr)
    if prof:
        cmd = [cmd[0], '-m', 'cProfile', '-o', os.path.join(td, 'cli.prof')] + cmd[1:]
    trace = os.environ.get('BBX_CLI_TRACE')                       # (debug: the child under rocprofv3 --kernel-trace, output in this directory)
    if trace:
        cmd = ['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', trace, '-o', 'r', '--'] + cmd
    import tempfile

    def cpu_stat():
        """the cgroup's CPU accounting (usage and throttling: a box gives this run a quota of cores)"""
        try:
            return {k: int(v) for k, v in (ln.split() for ln in open('/sys/fs/cgroup/cpu.stat'))}
        except (OSError, ValueError):
            return {}

    def one_run(out_dir):
        c0 = cpu_stat()
        t0 = time.time()
        r = subprocess.run(cmd + ['--red_dir', out_dir], env=dict(os.environ, BBX_TIMING='1'), capture_output=True, text=True, timeout=540)
        wall = time.time() - t0
        try:
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('BBX_TIMING ')]
            if r.returncode != 0 or not line:
                return dict(error='exit code %d' % r.returncode, stderr=r.stderr[-800:])
            tm = json.loads(line[-1][len('BBX_TIMING '):])
            done = tm.get('files_done_unix', [])
            marks = dict(tm['marks'])
            names = os.listdir(out_dir)
            nout = len([f_ for f_ in names if f_.endswith('_red.fits.fz')])
            mb = sum(os.path.getsize(os.path.join(out_dir, f_)) for f_ in names) / 1e6
            skip = min(16, max(1, len(done) // 3))
            d = dict(dir=os.path.dirname(out_dir), files=nfiles, products_of=nout, files_per_frame=round(len(names) / max(1, nout), 1),
                     MB_per_frame=round(mb / max(1, nout), 1), pipeline=tm.get('pipeline'), process_wall_s=round(wall, 2),
                     hbm_peak_GB_tensors=tm.get('hbm_peak_GB_tensors'),
                     seconds_before_the_list=round(marks.get('calibration_and_reference_files_in_hbm', 0.0), 2))
            c1 = cpu_stat()
            if c0 and c1:
                d['cgroup_cpu'] = dict(cores_used=round((c1.get('usage_usec', 0) - c0.get('usage_usec', 0)) / 1e6 / wall, 1),
                                       throttled_s=round((c1.get('throttled_usec', 0) - c0.get('throttled_usec', 0)) / 1e6, 2),
                                       periods_throttled=c1.get('nr_throttled', 0) - c0.get('nr_throttled', 0))
            if len(done) > skip + 1:
                d['frames_per_s'] = (len(done) - 1 - skip) / (done[-1] - done[skip])
                d['ms_per_frame'] = 1e3 / d['frames_per_s']
                d['frames_per_s_whole_list'] = len(done) / (done[-1] - (tm['t_module_import_unix'] + marks.get('calibration_and_reference_files_in_hbm', 0.0)))
                d['frames_per_s_process'] = len(done) / wall
                d['steady_state_from_file'] = skip
                t_list = tm['t_module_import_unix'] + marks.get('calibration_and_reference_files_in_hbm', 0.0)
                d['seconds_to_first_product'] = round(done[0] - t_list, 2)
                d['seconds_to_product_%d' % skip] = round(done[skip] - t_list, 2)
            return d
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    def scratch_run():
        # the same list with the products on local scratch (the raw files stay on the RAM disk)
        sd = tempfile.mkdtemp(prefix='bbx_list_out_', dir=tempfile.gettempdir())
        try:
            if shutil.disk_usage(sd).free > nfiles * 400e6 + 2e9:
                return one_run(os.path.join(sd, 'out_list'))
            return dict(skipped='not enough room in %s' % sd)
        finally:
            shutil.rmtree(sd, ignore_errors=True)
    both = not (prof or trace) and not os.environ.get('BBX_CLI_NO_POOL')
    res = {}
    if both and nfiles >= 48:
        # untimed warm-up, like the W warm-up steps of the headline: the same command over the first 48 files (a fresh box
        # gives its first list run 50-60 % of the rate of its third: page cache of the interpreter's modules, first-touch
        # pages of the RAM disk and of the pinned buffers)
        lstw = os.path.join(td, 'list_warm.txt')
        with open(lstw, 'w') as f:
            f.write('\n'.join(files[:48]) + '\n')
        cw = [c for c in cmd]
        cw[cw.index('--image_list') + 1] = lstw
        outw = os.path.join(td, 'out_warm')
        try:
            subprocess.run(cw + ['--red_dir', outw], env=dict(os.environ), capture_output=True, text=True, timeout=300)
        finally:
            shutil.rmtree(outw, ignore_errors=True)
        res['warmup_files'] = 48
        time.sleep(float(os.environ.get('BBX_CLI_SETTLE_S', '0')))     # (debug: idle seconds between the warm-up's exit and the measured runs)
    if both and os.environ.get('BBX_CLI_SCRATCH_FIRST'):          # (debug: the order of the two runs)
        res['scratch'] = scratch_run()
    res['ramdisk'] = one_run(os.path.join(td, 'out_list'))
    res['hbm_free_GB_before_the_children'] = round(free_b / 1e9, 1)                 # (what this bench process still holds is the rest of the card)
    if prof and os.path.isfile(os.path.join(td, 'cli.prof')):
        import pstats
        pstats.Stats(os.path.join(td, 'cli.prof'), stream=sys.stderr).sort_stats('cumulative').print_stats(45)
        pstats.Stats(os.path.join(td, 'cli.prof'), stream=sys.stderr).sort_stats('tottime').print_stats(40)
    if both and 'scratch' not in res:
        res['scratch'] = scratch_run()
    res['note'] = ('files to files through the operator\'s own entry: a child `python blackbox.py --image_list L ... --fpack True` over %d full-size '
                   'fpacked raw frames on the RAM disk; every product of every frame written (_red, _mask, _D, _Scorr, _Fpsf, _trans_limmag '
                   'tile-compressed on the lane that made them and written as .fits.fz by writer threads; _bkg_mini, _bkg_std_mini, _cat, '
                   '_trans and the header files by a worker process of the host pool); frames_per_s = steady state between the completion of '
                   'file [steady_state_from_file] and the last one, frames_per_s_whole_list from the first input to the last product, '
                   'frames_per_s_process over the process\'s wall time (imports, GPU context, masters into HBM included)' % nfiles)
    # the reference's own farm on one GPU (blackbox.py:363-379: pool_func(try_blackbox_reduce, files, nproc)): persistent
    # worker processes, a GPU context and the masters in HBM each, one file at a time per worker
    try:
        if os.environ.get('BBX_CLI_NO_POOL'):                       # (debug runs of the list alone)
            raise RuntimeError('skipped (BBX_CLI_NO_POOL)')
        npool, nf = 4, min(nfiles, 32)
        lst2 = os.path.join(td, 'list_pool.txt')
        with open(lst2, 'w') as f:
            f.write('\n'.join(files[:nf]) + '\n')
        cmd2 = [c for c in cmd1]
        i = cmd2.index('--image')
        cmd2[i:i + 2] = ['--image_list', lst2, '--nproc', str(npool)]
        out2 = os.path.join(td, 'out_pool')
        t0 = time.time()
        r2 = subprocess.run(cmd2 + ['--red_dir', out2], env=dict(os.environ, BBX_TIMING='1'), capture_output=True, text=True, timeout=540)
        wall2 = time.time() - t0
        n2 = len([f_ for f_ in os.listdir(out2) if f_.endswith('_red.fits.fz')]) if os.path.isdir(out2) else 0
        shutil.rmtree(out2, ignore_errors=True)
        res['process_pool'] = dict(workers=npool, files=nf, products_of=n2, wall_s=round(wall2, 2), frames_per_s=round(n2 / wall2, 2),
                                   returncode=r2.returncode,
                                   note='child `python blackbox.py --image_list L --nproc %d`: a spawn pool of persistent workers, each with its '
                                        'own GPU context and masters, files one by one per worker; wall time of the whole process '
                                        '(interpreter, pool start, %d x [torch import + context + masters into HBM] included)' % (npool, npool))
        if r2.returncode != 0:
            res['process_pool']['stderr'] = r2.stderr[-600:]
    except Exception as e:
        res['process_pool'] = dict(error=repr(e))
    return res


def process_cold(torch, ctx, raw, flat, bpm, ref, ref_mask, coeffs, sub_kw, box, raws=None, list_frames=0):
    """what ONE `python blackbox.py --image F` process costs at full size (the reference's Slurm contract: one interpreter per
    file, blackbox_slurm_google.py:305-309): the frame's inputs are written as files (fpacked raw, master flat, bad-pixel
    mask, crosstalk table, reference image + mask + sigma mini image, PSF stamp cubes), the command line runs as a child
    process twice (page cache cold-ish, then warm) with --fpack True, and reports where its wall time went (BBX_TIMING)"""
    import shutil
    import subprocess
    import tempfile
    from blackbox_amd import fitsio
    from blackbox_amd import fpack as P
    td = tempfile.mkdtemp(prefix='bbx_proc_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    try:
        hdr = {'DATE-OBS': '2024-01-02T03:04:05', 'EXPTIME': 60.0, 'IMAGETYP': 'object', 'FILTER': 'q', 'OBJECT': 'synthetic'}
        rawf = P.fpack_image(ctx, os.path.join(td, 'ML1_raw.fits'), raw, hdr)
        fitsio.write_image(os.path.join(td, 'flat.fits'), flat.cpu().numpy())
        fitsio.write_image(os.path.join(td, 'bpm.fits'), bpm.cpu().numpy())
        fitsio.write_image(os.path.join(td, 'ref.fits'), ref.cpu().numpy())
        fitsio.write_image(os.path.join(td, 'ref_mask.fits'), ref_mask.cpu().numpy())
        fitsio.write_image(os.path.join(td, 'ref_std_mini.fits'), np.asarray(sub_kw['ref_bkg_std_mini'], np.float32))
        fitsio.write_image(os.path.join(td, 'psf_new.fits'), sub_kw['psf_new'].cpu().numpy())
        fitsio.write_image(os.path.join(td, 'psf_ref.fits'), sub_kw['psf_ref'].cpu().numpy())
        with open(os.path.join(td, 'xtalk.dat'), 'w') as f:
            f.write('victim source correction\n')
            for s_ in range(16):
                for v_ in range(16):
                    if s_ != v_:
                        f.write('%d %d %.10e\n' % (v_ + 1, s_ + 1, coeffs[s_, v_]))
        cmd = [sys.executable, os.path.join(ROOT, 'blackbox.py'), '--telescope', 'ML1', '--img_reduce', 'True', '--cat_extract', 'True',
               '--trans_extract', 'True', '--force_reproc_new', 'True', '--image', rawf, '--mflat', os.path.join(td, 'flat.fits'),
               '--bpm', os.path.join(td, 'bpm.fits'), '--crosstalk', os.path.join(td, 'xtalk.dat'), '--ref', os.path.join(td, 'ref.fits'),
               '--ref_mask', os.path.join(td, 'ref_mask.fits'), '--ref_bkg_std_mini', os.path.join(td, 'ref_std_mini.fits'),
               '--ref_bkgsub', 'True', '--psf_new', os.path.join(td, 'psf_new.fits'), '--psf_ref', os.path.join(td, 'psf_ref.fits'),
               '--fratio', '1.0', '--zogy_dx', '0.03', '--zogy_dy', '0.03', '--fpack', 'True', '--bkg_boxsize', str(box)]
        runs = []
        for k in range(2):
            out_dir = os.path.join(td, 'out%d' % k)
            t0 = time.time()
            r = subprocess.run(cmd + ['--red_dir', out_dir], env=dict(os.environ, BBX_TIMING='1'), capture_output=True, text=True, timeout=280)
            wall = time.time() - t0
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('BBX_TIMING ')]
            if r.returncode != 0 or not line:
                return dict(error='exit code %d' % r.returncode, stderr=r.stderr[-600:])
            tm = json.loads(line[-1][len('BBX_TIMING '):])
            marks = tm['marks']
            d = {}
            prev = 0.0
            for name, t in marks:
                d[name] = round(t - prev, 3)
                prev = t
            files = sorted(os.listdir(out_dir))
            runs.append(dict(wall_s=round(wall, 3), interpreter_start_s=round(tm['t_module_import_unix'] - t0, 3), seconds_per_phase=d,
                             files_written=len(files), MB_written=round(sum(os.path.getsize(os.path.join(out_dir, f_)) for f_ in files) / 1e6, 1)))
        res = dict(first_run=runs[0], second_run=runs[1],
                   note='child process `python blackbox.py --image raw.fits.fz ... --fpack True` at full size, all products of the frame '
                        'written; seconds_per_phase = time between consecutive marks inside the process (imports, GPU context, masters / '
                        'reference / PSFs into HBM, raw read + decode, reduction, subtraction, writing)')
        if list_frames:
            res['image_list'] = cli_image_list(ctx, td, cmd, raws or [raw], hdr, list_frames)
        return res
    except Exception as e:
        return dict(error=repr(e))
    finally:
        shutil.rmtree(td, ignore_errors=True)


def load_pmc(args):
    """HBM traffic per launch from the tracked PMC summary (profiles/*_pmc_traffic.json: two --pmc passes of this
    command, FETCH_SIZE / WRITE_SIZE with the gfx950 corrections) -- only while the kernels it was taken from are
    the ones in the tree: the summary records the SHA-256 of the HIP sources it profiled"""
    import glob
    import hashlib
    if args.small or args.raw != 'u16' or args.workload != 'zogy':
        return None, 'no PMC summary for this configuration'
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files:
        return None, 'no PMC summary'
    path = files[-1]
    try:
        j = json.load(open(path))
        want = j.get('source_sha256', {})
        if not want:
            return None, '%s carries no source stamp: not used' % os.path.relpath(path, ROOT)
        for rel, sha in want.items():
            if hashlib.sha256(open(os.path.join(ROOT, rel), 'rb').read()).hexdigest() != sha:
                return None, '%s is stale (%s changed since the PMC run): traffic not reported' % (os.path.relpath(path, ROOT), rel)
        return j['kernels'], '%s (commit %s)' % (os.path.relpath(path, ROOT), j.get('commit', '?'))
    except Exception as e:
        return None, 'PMC summary unreadable: %s' % e


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child
    launcher (fresh processes; nothing in this process has touched the GPU) and pass its exit code on"""
    import subprocess
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='zogy', choices=['zogy', 'full', 'calib'],
                    help='zogy: reduce + ZOGY + photometry (headline, configs[4] per frame); full: configs[2]; calib: configs[1]')
    ap.add_argument('--raw', default='u16', choices=['u16', 'f32'])
    ap.add_argument('--small', action='store_true', help='reduced geometry (debug)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the other workloads and the I/O-inclusive figures')
    ap.add_argument('--depth', type=int, default=None, help='frames in flight')
    ap.add_argument('--workers', type=int, default=None, help='host fit worker processes')
    ap.add_argument('--lanes', type=int, default=None, help='stage-C lanes (context + stream + issuing thread) per GPU')
    ap.add_argument('--writers', type=int, default=12, help='writer threads of the output stage (io_inclusive.measured)')
    ap.add_argument('--readers', type=int, default=4, help='reader threads of the input stage (io_inclusive.measured)')
    ap.add_argument('--io-only', action='store_true', help='only the measured I/O-inclusive run (debug)')
    ap.add_argument('--proc-only', action='store_true', help='only the per-file process timing (debug)')
    ap.add_argument('--io-simple', action='store_true', help='with --io-only: one files-to-files run on the RAM disk (profiling)')
    ap.add_argument('--psf-size', type=int, default=49, help='side of the PSF stamps of the ZOGY stage (SURVEY 8d: 49)')
    args = ap.parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    from blackbox_amd import reduce as R

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    # rehearsal knobs (a 1-GPU box): BBX_BENCH_BACKEND=gloo BBX_BENCH_ONE_GPU=1 run all ranks on cuda:0
    backend = os.environ.get('BBX_BENCH_BACKEND', 'nccl')
    one_gpu = bool(os.environ.get('BBX_BENCH_ONE_GPU'))
    if one_gpu:
        local = 0
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    # the fit workers are started before this process makes its first GPU call
    from blackbox_amd.pipeline import HostPool
    pool = HostPool(args.workers)
    ctx = R.Context(local)
    dev = ctx.device
    if args.small:
        ysz, xsz, os_y, os_x = 660, 330, 20, 45
        size, border, box = 330, 20, 30
    else:
        ysz, xsz, os_y, os_x = 5280, 1320, 20, 180
        size, border, box = 1320, 40, 60
    wl = args.workload
    lanes = args.lanes or {'zogy': 6, 'full': 6, 'calib': 8}[wl]          # (calib: 4 / 6 / 8 / 10 lanes -> 1180 / 1230 / 1370 / 1210 frames/s)
    depth = args.depth or {'zogy': 16, 'full': 18, 'calib': 24}[wl]
    seed = 1000 * 4 + rank
    # the scene follows the PSF model ZOGY is handed (zogy_psf_field): star shapes per sub-image, reference fluxes / fratio
    raw, flat, bpm, ex = synth_frame_device(torch, dev, ysz, xsz, os_y, os_x, seed, args.raw, extras=True, ntrans=50,
                                            psf_field=zogy_psf_field(2 * ysz // size, 8 * xsz // size, size))
    ref, ref_mask = synth_reference(torch, dev, ex.pop('scene0'), seed)
    geom = R.geometry(raw.shape, ysz, xsz)
    tel = 'ML1'
    cpu_full = None
    if rank == 0 and world == 1 and not args.no_cpu and not args.small and args.raw == 'u16' and not (args.io_only or args.proc_only):
        # the frame of this run for the CPU baseline's whole-frame check (host copies on the RAM disk, removed after use)
        cpu_full = os.path.join('/dev/shm' if os.path.isdir('/dev/shm') else '/tmp', 'bbx_cpu_full_%d_' % os.getpid())
        np.save(cpu_full + 'raw.npy', raw.cpu().numpy()); np.save(cpu_full + 'flat.npy', flat.cpu().numpy()); np.save(cpu_full + 'bpm.npy', bpm.cpu().numpy())
    N = 2 * ysz * 8 * xsz
    # distinct raw buffers for the frames in flight (same scene, fresh read noise): the working set
    # of consecutive frames must not sit in the 256 MB Infinity Cache
    nbuf = max(depth, 4)
    g = torch.Generator(device=dev); g.manual_seed(seed + 1)
    raws = [raw]
    for _ in range(nbuf - 1):
        r = (raw.to(torch.float32) + 2.0 * torch.randn(raw.shape, device=dev, generator=g)).round().clamp(0, 65535)
        raws.append(r.to(torch.int32).to(torch.uint16).contiguous() if args.raw == 'u16' else r.contiguous())
    rs = np.random.RandomState(0)
    coeffs = np.zeros((16, 16)); coeffs[~np.eye(16, dtype=bool)] = rs.uniform(0, 2e-4, 240)
    S = args.psf_size
    sub_kw = zogy_inputs(torch, dev, 2 * ysz // size, 8 * xsz // size, S, box, 2 * ysz, 8 * xsz)
    sub_kw.update(ref=ref, ref_mask=ref_mask, cat_extract=True, trans_extract=True, subimage_size=size, subimage_border=border,
                  bkg_boxsize=box)
    base_kw = dict(mflat=flat, bpm=bpm)
    kws = {
        'calib': dict(base_kw),
        'full': dict(base_kw, xtalk_coeffs=coeffs, do_finish=True, detect_sats=True,
                     subtract=dict(psf_new=None, psf_ref=None, ref=None, ref_mask=None, trans_extract=False,
                                   subimage_size=size, subimage_border=border, bkg_boxsize=box)),
        'zogy': dict(base_kw, xtalk_coeffs=coeffs, do_finish=True, detect_sats=True, subtract=sub_kw),
    }

    import ctypes as C
    from blackbox_amd import _lib

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.proc_only:
        print(json.dumps(process_cold(torch, ctx, raw, flat, bpm, ref, ref_mask, coeffs, sub_kw, box, raws=raws, list_frames=args.steps if args.steps != 60 else 96)))
        pool.close()
        return
    if args.io_only:
        def section(name):
            sys.stderr.write('[bench] %s\n' % name); sys.stderr.flush()
        print(json.dumps(measure_io(torch, ctx, tel, geom, raws, kws[wl], depth, lanes, pool, barrier, args, section, frames=args.steps)))
        pool.close()
        return
    # ---- the child-process measurements first: `python blackbox.py` per file, over a list (files to files) and as a pool.
    # They are the deployment -- no bench process works the GPU beside them there --, so they run while this process has
    # done nothing on the GPU but make the synthetic inputs: started behind its own pipelines (which leave their queues,
    # workspaces and a busy card behind) the first list run measured 30-49 frames/s in four of seven default runs, 65-79 here.
    early = {}
    if rank == 0 and world == 1 and not args.no_extras and wl == 'zogy' and not args.small:
        t_e = time.perf_counter()
        early['process_per_file'] = process_cold(torch, ctx, raw, flat, bpm, ref, ref_mask, coeffs, sub_kw, box, raws=raws, list_frames=96)
        sys.stderr.write('[bench] process_per_file + image_list: %.1f s\n' % (time.perf_counter() - t_e)); sys.stderr.flush()
    # ---- serial reference run of one frame (stage breakdown + isolated kernel timings; untimed) --
    from blackbox_amd import zogy as G
    stage_ms = {}

    def frame_serial(with_sub):
        header, hm = {}, {}
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
        ev[0].record()
        R.gain_corr(header, tel)
        sol = R.os_solve(ctx, raw, header, tel, geom)
        ev[1].record()
        data, mask = R.calibrate(ctx, raw, sol, header, hm, tel, geom, mflat=flat, bpm=bpm)
        ev[2].record()
        R.mask_init_finish(ctx, mask, header, hm, geom)
        ev[3].record()
        st = R.cosmics_corr(ctx, data, header, mask, hm, tel)
        ev[4].record()
        R.xtalk_corr(ctx, data, coeffs, mask, geom)
        R.sat_detect(ctx, data, header, mask, hm)
        R.mask_header(ctx, mask, hm)
        R.edge_fill(ctx, data, mask, geom)
        ev[5].record()
        res = None
        if with_sub:
            res = G.optimal_subtraction(ctx, data, new_mask=mask, **sub_kw)
        ev[6].record()
        ctx.sync()
        return ev, st, res

    frame_serial(wl == 'zogy')
    _lib.check(_lib.lib.bbx_profile_enable(ctx.h, 1), 'bbx_profile_enable')
    nser = 3
    for _ in range(nser):
        frame_serial(wl == 'zogy')
    iso_ms = (C.c_double * NSL)()
    iso_calls = (C.c_int32 * NSL)()
    _lib.check(_lib.lib.bbx_profile_read(ctx.h, iso_ms, iso_calls, NSL), 'bbx_profile_read', ctx.h)
    _lib.check(_lib.lib.bbx_profile_enable(ctx.h, 0), 'bbx_profile_enable')
    t0 = time.perf_counter()
    ev, st, res = frame_serial(wl == 'zogy')
    latency_ms = 1e3 * (time.perf_counter() - t0)
    for i, n in enumerate(['overscan(stats+host fits)', 'calibrate', 'mask_finish', 'lacosmic', 'xtalk+sat+counts+edge_fill',
                           'optimal_subtraction']):
        stage_ms[n] = ev[i].elapsed_time(ev[i + 1])
    stats = st.cpu().numpy().tolist()
    sub_info = None
    if res is not None:
        sub_info = dict(ntransients=len(res['transients']), ncatalog=int(len(res['catalog']['X_POS'])) if res['catalog'] else 0,
                        scorr_median=res['header_trans']['Z-SCMED'][0], scorr_std=res['header_trans']['Z-SCSTD'][0],
                        injected=len(ex['transients']))
    del res
    torch.cuda.empty_cache()

    # ---- timed region ----------------------------------------------------------------------------
    r = run_pipeline(torch, ctx, tel, geom, raws, kws[wl], args.steps, args.warmup, depth, lanes, pool, barrier, prof_ctx=ctx)
    dt, dt_all = r['dt'], r['dt_all']
    ms_tot, calls = r['prof']
    if world > 1:
        t = torch.tensor([dt, dt_all], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, dt_all = float(t[0].item()), float(t[1].item())

    out = None
    if rank == 0:
        b_raw = 2 if args.raw == 'u16' else 4
        L = size + 2 * border
        nsub = (2 * ysz // size) * (8 * xsz // size)
        # algorithmic bytes per launch
        HP = ((L // 2 + 1 + 3) // 4) * 4                     # padded half-spectrum width of bbx_zogy3.hip (NL = 4)
        spec = nsub * HP * L * 8                              # one half spectrum of all sub-images, bytes
        cut = 4 * nsub * L * L                                # one frame read as overlapping sub-images, bytes
        rows_once = not (calls[10] and calls[11]) or calls[10] < 2 * calls[11]
        wfrac = min(1.0, (4 * S + 32) / float(L))             # row window of the matched-filter kernels
        kern = {
            'k_calibrate': (0, b_raw * N + 4 * N + N + 4 * N + N),
            'k_lac_cand': (1, 4 * N),
            # the kernels of bbx_zogy_frame, half-spectrum arrays of [spec] bytes:
            'k_final_rows': (7, 4 * spec + 4 * 4 * N),          # reads D^, V_S^, S_n^, S_r^ (column-transformed), writes D, Scorr, Fpsf, Fpsferr
            'k_psf_rowdft': (13, int(2 * nsub * S * (S * 4 + HP * 8))),        # the stamps in, their row DFTs out
            'k_psf_cols': (8, int((2 + 2 * wfrac) * spec)),     # writes the two PSF spectra (one float4 array) and the window rows of the two column-inverted k^
            'k_psf_rows': (9, int(4 * wfrac * spec)),           # the window rows of k_r, k_n in, their squares' row pass out
            # one launch for both pairs (4 frame cuts read, 4 half spectra written); frames whose rows are not 16-byte aligned
            # take two launches (2 + 4 cuts read, 2 spectra written each: the average below)
            # (round 5: the sigma maps come off their mini images, bbx_zogy_frame_mini: 2 frame cuts read instead of 4)
            'k_img_rows': (10, int(4 * spec + 2 * cut) if rows_once else int(2 * spec + 3 * cut)),
            'k_img_cols': (11, int(7 * spec)),                  # N^, R^, (Pn^, Pr^) in; D^, S_n^, S_r^ out
            'k_var_cols': (12, int((3 + 2 * wfrac) * spec)),    # V_n^, V_r^ and the window rows of (k^2)^ in, V_S^ out
        }
        zogy_kernels = ('k_psf_rowdft', 'k_psf_cols', 'k_psf_rows', 'k_img_rows', 'k_img_cols', 'k_var_cols', 'k_final_rows')
        zogy_io_model = int(4 * 4 * nsub * L * L + 4 * 4 * N)     # 4 inputs read with the tile overlap, 4 outputs written
        iso = {k: (iso_ms[sl] / max(1, iso_calls[sl]), by, iso_calls[sl]) for k, (sl, by) in kern.items() if iso_calls[sl]}
        live = {k: (ms_tot[sl] / max(1, calls[sl]), by, calls[sl]) for k, (sl, by) in kern.items() if calls[sl]}

        def gbs(t):
            return t[1] / (t[0] * 1e-3) / 1e9

        def kdict(t):
            return dict(avg_launch_ms=t[0], launches=int(t[2]), moved_bytes_per_launch=int(t[1]), achieved_moved=gbs(t),
                        frac_moved=gbs(t) / HBM_PEAK_GBS)
        pmc, pmc_note = load_pmc(args)
        per_frame = {k: (2 if k == 'k_img_rows' and not rows_once else 1) for k in zogy_kernels}
        # SURVEY 8d frame figure (config 5, u16 raw): calibration 12N(u16)/14N(f32) + LA-Cosmic 22N + xtalk 9N + masks 6N +
        # sat 8N + mesh 9N + ZOGY 34N + photometry (n_src x S^2 x 3 images x 4 B)
        Nraw = raw.numel()
        frame_bytes = int((b_raw * Nraw + 10 * N) + 22 * N)                                       # configs[1]
        if wl != 'calib':
            frame_bytes += int(9 * N + 6 * N + 8 * N + 9 * N)                                     # configs[2]
        if wl == 'zogy':
            frame_bytes += int(zogy_io_model + 12000 * S * S * 12)                                # configs[4]
        if wl == 'zogy' and all(k in live for k in zogy_kernels) and all(k in iso for k in zogy_kernels):
            # the dominant launch group of the headline workload: bbx_zogy_frame = 6 launches per frame (one library call).
            # achieved = SURVEY 8d algorithmic bytes of the stage (inputs once with the tile overlap + outputs once)
            # / the summed duration of its launches in the timed region (HIP events stamped by the launches themselves,
            # lane 0's stream)
            zms = sum(live[k][0] * per_frame[k] for k in zogy_kernels)
            zms_iso = sum(iso[k][0] * per_frame[k] for k in zogy_kernels)
            moved = int(sum(live[k][1] * per_frame[k] for k in zogy_kernels))
            roof = dict(bound='hbm', kernel='bbx_zogy_frame (launch group: k_psf_rowdft, k_psf_cols, k_psf_rows, k_img_rows, k_img_cols, '
                                            'k_var_cols, k_final_rows)',
                        achieved=zogy_io_model / (zms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s',
                        avg_launch_ms=zms, launches=int(min(live[k][2] // per_frame[k] for k in zogy_kernels)),
                        bytes_per_launch=zogy_io_model, traffic=None,
                        bytes_model='SURVEY 8d: 4 inputs (new, ref, 2 sigma images) read once with the tile overlap (L/size)^2 + '
                                    '4 outputs (D, Scorr, Fpsf, Fpsferr) written once = 34N',
                        alone=dict(avg_launch_ms=zms_iso, achieved=zogy_io_model / (zms_iso * 1e-3) / 1e9,
                                   frac=zogy_io_model / (zms_iso * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   note='serial frames before the timed region: the group alone on the GPU'),
                        moved_bytes_by_design=moved, achieved_moved=moved / (zms * 1e-3) / 1e9,
                        frac_moved=moved / (zms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        kernels={k: dict(kdict(live[k]), alone_ms=iso[k][0], per_frame=per_frame[k]) for k in zogy_kernels},
                        others={k: kdict(live[k]) for k in live if k not in zogy_kernels})
            if pmc is not None and all(k in pmc for k in zogy_kernels):
                roof['traffic'] = int(sum(pmc[k]['traffic_bytes_per_launch'] * per_frame[k] for k in zogy_kernels))
                roof['traffic_per_kernel'] = {k: pmc[k]['traffic_bytes_per_launch'] for k in zogy_kernels}
        else:
            dom = 'k_calibrate' if 'k_calibrate' in live else max(live, key=lambda k: live[k][0])
            roof = dict(bound='hbm', kernel=dom, achieved=gbs(live[dom]), peak=HBM_PEAK_GBS, unit='GB/s',
                        avg_launch_ms=live[dom][0], launches=int(live[dom][2]), bytes_per_launch=live[dom][1], traffic=None,
                        others={k: kdict(live[k]) for k in live if k != dom})
            if dom in iso:
                roof['alone'] = dict(avg_launch_ms=iso[dom][0], achieved=gbs(iso[dom]), frac=gbs(iso[dom]) / HBM_PEAK_GBS)
            if pmc is not None and dom in pmc:
                roof['traffic'] = pmc[dom]['traffic_bytes_per_launch']
        roof['frac'] = roof['achieved'] / roof['peak']
        roof['traffic_source'] = pmc_note
        roof['timing'] = ('HIP events stamped by the launch itself (hipExtLaunchKernelGGL start / stop events: the kernel\'s execution) '
                          'on the launch streams, over the timed region, all %d lanes (kernels of different lanes run concurrently)' % lanes)
        roof['frame'] = dict(bytes=frame_bytes, achieved=frame_bytes / (dt / args.steps) / 1e9,
                             frac=frame_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                             note='SURVEY 8d algorithmic bytes of a whole frame of this workload (config 5: 11.45 GB with a u16 raw) / ms_per_step')
        shape = '%dx%d raw (%s)' % (raw.shape[0], raw.shape[1], args.raw)
        names = {'zogy': 'configs[4] per frame on one GPU: one %s -> reduce (gain+overscan+flat+mask+LA-Cosmic(niter=3)+xtalk+'
                         'sat trail+counts+edge fill) + optimal_subtraction vs a co-added, background-subtracted reference with its '
                         'bkg_std_mini (buildref products): bkg mesh of the new frame, variance images, %d '
                         'sub-images of %d^2 ZOGY with %d distinct %dx%d Moffat PSF pairs and per-sub-image flux ratio / sigma / dx / dy, '
                         'D/Scorr/Fpsf/Fpsferr, transients, PSF-photometry catalogue), ML1'
                         % (shape, nsub, L, nsub, S, S),
                 'full': 'configs[2]: one %s -> full calibration + LA-Cosmic + xtalk + sat trail + counts + edge fill + '
                         'background mesh and subtraction, ML1' % shape,
                 'calib': 'configs[1]: one %s -> gain+overscan+flat+mask_init+LA-Cosmic(niter=3), ML1' % shape}
        out = dict(metric='10560x10560 fp32 frames/sec end-to-end reduce+ZOGY; % HBM roofline' if wl == 'zogy' else
                   '10560x10560 fp32 frames/sec end-to-end reduce (%s)' % wl,
                   value=args.steps * world / dt, unit='frames/s', n_gpus=1 if one_gpu else world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True,
                   scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload=names[wl],
                               frames_per_gpu=args.steps, frames_in_flight=depth, stageC_lanes=lanes, host_fit_workers=r['nworkers'],
                               distinct_raw_buffers=nbuf, parallelism='frame-per-gpu x%d (no collective)' % world,
                               **(dict(rehearsal_one_gpu=True, ranks=world,
                                       rehearsal_note='BBX_BENCH_ONE_GPU: all %d ranks share cuda:0 -- a rehearsal of the multi-rank '
                                                      'path on a one-GPU box, NOT a scaling measurement' % world) if one_gpu else {})),
                   timing='steady state: %d frames fill the pipeline, %d warm-up frames, then the clock runs from the completion of '
                          'the last warm-up frame to the completion of the %d-th frame after it, with %d more frames in flight '
                          'behind it (no fill, no drain inside the region); barrier + device synchronisation around the run'
                          % (depth, args.warmup, args.steps, depth),
                   idle_to_idle=dict(frames=r['n_all'], frames_per_s=r['n_all'] * world / dt_all,
                                     note='all frames of the run from an idle device to an idle device (pipeline fill and drain included)'),
                   pipeline_wall_ms_per_frame=dict(zip(['stageA_stats', 'stageB_host_fits', 'stageC_device'],
                                                       [1e3 * t / max(1, r['t_stats'][3]) for t in r['t_stats'][:3]])),
                   host_ms_per_frame=r['host_ms_per_frame'],
                   single_frame_latency_ms=latency_ms, stage_ms_serial=stage_ms, lacosmic_stats=stats, subtraction=sub_info,
                   roofline=roof)
    # ---- the other workloads, briefly, and the I/O-inclusive figures (not part of `value`) ---------
    t_sec = [time.perf_counter()]

    def section(name):
        """progress on stderr: the extras below take most of the run's wall time"""
        now = time.perf_counter()
        sys.stderr.write('[bench] %s: %.1f s\n' % (name, now - t_sec[0]))
        sys.stderr.flush()
        t_sec[0] = now
    if world == 1 and not args.no_extras:
        # the same steady-state measurement over a long run (the headline's K frames are few), first of the extras: behind
        # the other workloads' pipelines (more lanes, more frames in flight) the same 240 frames measure 8-10 % lower
        r3 = run_pipeline(torch, ctx, tel, geom, raws, kws[wl], 240, 4, depth, lanes, pool, barrier)
        out['long_run'] = dict(frames=240, frames_per_s=240 / r3['dt'], ms_per_frame=1e3 * r3['dt'] / 240)
        section('long_run')
        others = {}
        for w2 in ('calib', 'full'):
            if w2 == wl:
                continue
            l2, d2 = (8, 48) if w2 == 'calib' else (6, 18)      # (calib: a frame takes ~20 ms from its strip statistics to its last kernel; 24 in flight cap the rate near 1100)
            r2 = run_pipeline(torch, ctx, tel, geom, raws, kws[w2], 60, 4, d2, l2, pool, barrier)
            others[w2] = dict(frames_per_s=60 / r2['dt'], ms_per_frame=1e3 * r2['dt'] / 60, frames_in_flight=d2, lanes=l2)
            section('workload ' + w2)
        if wl == 'zogy':
            # the same workload against a reference that still carries a sky of its own: its background mesh and
            # sigma image are then made per frame as well (bkg mesh x2)
            ref2 = ref + 120.0
            kw2 = dict(kws['zogy'], subtract=dict(sub_kw, ref=ref2, ref_is_bkgsub=False, ref_bkg_std_mini=None))
            r2 = run_pipeline(torch, ctx, tel, geom, raws, kw2, 30, 4, depth, lanes, pool, barrier)
            others['zogy_ref_with_sky'] = dict(frames_per_s=30 / r2['dt'], ms_per_frame=1e3 * r2['dt'] / 30, frames_in_flight=depth,
                                               lanes=lanes, note='reference not background-subtracted: bkg mesh x2 per frame')
            del ref2
            section('zogy_ref_with_sky')
            # (rounds 1-4 also timed one 25 x 25 stamp for all sub-images here: the scene now follows the PSF field that ZOGY is
            # handed, a single stamp no longer matches its stars -- the stage's dependence on S is in HISTORY.md: 0.05 ms)
        out['other_workloads'] = others
        # the stage figures next to the headline's roofline (north_star: >= 60 % of the HBM roofline on the calibration +
        # LA-Cosmic stage): frames/s of the stage workloads x their SURVEY 8d algorithmic bytes, and the launch-level
        # fractions of the two kernels that carry that stage, from the headline's timed region.  (8d prices LA-Cosmic at 22N;
        # the exact sparse formulation moves ~4N + lists -- the whole-frame bit-exact test proves the work is done -- so the
        # stage fraction is a statement about throughput in 8d's currency, the kernel fractions about the kernels.)
        b_raw2 = 2 if args.raw == 'u16' else 4
        by_calib = int(b_raw2 * raw.numel() + 10 * N + 22 * N)
        by_full = by_calib + int(9 * N + 6 * N + 8 * N + 9 * N)
        stages = {}
        for w2, by in (('calib', by_calib), ('full', by_full)):
            fps = others[w2]['frames_per_s'] if w2 in others else (out['value'] if w2 == wl else None)
            if fps:
                stages[w2] = dict(frames_per_s=fps, bytes_8d=by, achieved=by * fps / 1e9, frac=by * fps / 1e9 / HBM_PEAK_GBS, unit='GB/s')
        ko = out['roofline'].get('others', {})
        for k in ('k_calibrate', 'k_lac_cand'):
            if k in ko:
                stages.setdefault('kernels', {})[k] = dict(frac_moved=ko[k]['frac_moved'], avg_launch_ms=ko[k]['avg_launch_ms'])
        out['roofline']['stages'] = stages
        out['io_inclusive'] = io_inclusive(torch, ctx, raw, N, out['ms_per_step'], wl)
        section('io_inclusive (pcie, serial writers)')
        if 'process_per_file' in early:
            out['process_per_file'] = early['process_per_file']
            il = out['process_per_file'].pop('image_list', None)
            if il is not None:
                # files to files, measured through the operator's own entry: a child `python blackbox.py --image_list` (round 4
                # timed bench.py's own loop around the pipeline here; that loop is still there as `bench.py --io-only`)
                pp = il.pop('process_pool', None)
                out['io_inclusive']['measured'] = il
                if pp is not None:
                    out['process_pool'] = pp
        else:
            out['io_inclusive']['measured'] = measure_io(torch, ctx, tel, geom, raws, kws[wl], depth, lanes, pool, barrier, args, section)
    pool.close()
    if rank == 0:
        if not args.no_cpu and world == 1:                           # (the CPU baseline: at N = 1 only)
            full_path = None
            if not args.small and world == 1 and cpu_full is not None:
                full_path = cpu_full
            out['cpu_baseline'] = cpu_baseline(wl, full_path)
            if full_path:
                for k in ('raw', 'flat', 'bpm'):
                    try:
                        os.unlink(full_path + k + '.npy')
                    except OSError:
                        pass
            section('cpu_baseline')
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


def io_inclusive(torch, ctx, raw, N, ms_compute, wl):
    """SURVEY 8d timing items (ii) and (iii) as bounds from measured parts: PCIe copies of what a
    frame moves (pinned memory, H2D and D2H on their own streams) and, with the products
    tile-compressed on the device (bbx_fpack_tiles), the compressed bytes instead"""
    from blackbox_amd import fpack as P
    dev = ctx.device
    nprod = 5 if wl == 'zogy' else 1                            # float32 images leaving: red (+ D, Scorr, Fpsf, Fpsferr)
    h_raw = torch.empty(raw.shape, dtype=raw.dtype, pin_memory=True)
    d_raw = torch.empty_like(raw)                                 # (the upload target: the benchmark's raw frames stay as they are)
    h_img = torch.empty(N, dtype=torch.float32, pin_memory=True)
    h_msk = torch.empty(N, dtype=torch.uint8, pin_memory=True)
    d_img = torch.randn(N, device=dev)
    d_msk = torch.zeros(N, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        with torch.cuda.stream(s1):
            d_raw.copy_(h_raw, non_blocking=True)
        with torch.cuda.stream(s2):
            for _ in range(nprod):
                h_img.copy_(d_img, non_blocking=True)
            h_msk.copy_(d_msk, non_blocking=True)
    torch.cuda.synchronize()
    ms_pcie = 1e3 * (time.perf_counter() - t0) / reps
    out = dict(pcie_ms_per_frame=ms_pcie, bytes_h2d=raw.numel() * raw.element_size(), bytes_d2h=nprod * 4 * N + N,
               frames_per_s_pcie_bound=1e3 / max(ms_pcie, ms_compute),
               note='raw H2D + products D2H through pinned memory on two streams, overlapped with compute: '
                    'rate = 1 / max(compute, copies)')
    try:
        img = (250 + 20 * torch.randn(int(np.sqrt(N)), int(np.sqrt(N)), device=dev)).contiguous()
        P.compress_tiles(ctx, img, 16, 1, _view=True); ctx.sync()
        t0 = time.perf_counter()
        cd = P.compress_tiles(ctx, img, 16, 1, _view=True); ctx.sync()
        ms_fp = 1e3 * (time.perf_counter() - t0)
        out['with_fpack'] = dict(ms_per_float_image_incl_D2H=ms_fp, compressed_MB=cd['heap'].size / 1e6,
                                 frames_per_s_bound=1e3 / max(ms_compute, nprod * ms_fp, ms_pcie * (raw.numel() * raw.element_size()) /
                                                              (raw.numel() * raw.element_size() + nprod * 4 * N + N)),
                                 note='products quantised (q=16) + Rice-compressed on the device, only the streams cross PCIe')
    except Exception as e:                                         # the figure is informative only
        out['with_fpack'] = dict(error=str(e))
    # files included (SURVEY 8d iii): one frame's products written by the product's own FITS writers to a scratch
    # directory -- uncompressed float32 (D2H + fitsio.write_image) and as .fits.fz (fpack.fpack_image: compressed on the
    # device, the host writes the streams) -- one writer thread, as the per-file CLI does it
    try:
        import tempfile
        from blackbox_amd import fitsio
        side = int(np.sqrt(N))
        img2 = (250 + 20 * torch.randn(side, side, device=dev)).contiguous()
        with tempfile.TemporaryDirectory(prefix='bbx_bench_') as td:
            t0 = time.perf_counter()
            fitsio.write_image(os.path.join(td, 'a.fits'), img2.cpu().numpy())
            ms_plain = 1e3 * (time.perf_counter() - t0)
            P.fpack_image(ctx, os.path.join(td, 'warm.fits.fz'), img2)
            t0 = time.perf_counter()
            P.fpack_image(ctx, os.path.join(td, 'b.fits.fz'), img2)
            ms_fz = 1e3 * (time.perf_counter() - t0)
            sz_plain, sz_fz = os.path.getsize(os.path.join(td, 'a.fits')), os.path.getsize(os.path.join(td, 'b.fits.fz'))
        out['fits_inclusive'] = dict(ms_per_float_image_uncompressed=ms_plain, ms_per_float_image_fz=ms_fz, MB_uncompressed=sz_plain / 1e6,
                                     MB_fz=sz_fz / 1e6, float_images_per_frame=nprod,
                                     frames_per_s_one_writer_uncompressed=1e3 / (nprod * ms_plain),
                                     frames_per_s_one_writer_fz=1e3 / (nprod * ms_fz),
                                     note='one host thread writing the float products of a frame one after the other to local scratch '
                                          '(the mask and the tables are small beside them); writers of several frames run in parallel '
                                          'in a deployment (one per lane)')
    except Exception as e:
        out['fits_inclusive'] = dict(error=str(e))
    return out


if __name__ == '__main__':
    main()
