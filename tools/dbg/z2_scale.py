"""bbx_zogy_frame on frames of 4 .. 64 sub-images: time per sub-image (does a working set that fits the Infinity Cache help?)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
import bench
from blackbox_amd import reduce as R, zogy as G
ctx = R.Context(0)
dev = ctx.device
for (nsy, nsx) in ((1, 2), (2, 2), (2, 4), (4, 4), (4, 8), (8, 8)):
    ny, nx = nsy * 1320, nsx * 1320
    nsub = nsy * nsx
    g = torch.Generator(device=dev); g.manual_seed(1)
    new = (20 * torch.randn(ny, nx, device=dev, generator=g)).contiguous()
    ref = (8 * torch.randn(ny, nx, device=dev, generator=g)).contiguous()
    sn = torch.full((ny, nx), 20.0, device=dev); sr = torch.full((ny, nx), 8.0, device=dev)
    psf = torch.from_numpy(np.repeat(bench.moffat_stamp(25, 4.0)[None], nsub, 0)).to(dev)
    scal = np.tile(np.array([[20, 8, 1, 1, 0.03, 0.03]], np.float32), (nsub, 1))
    for rep in range(2):
        outs = G.run_zogy_frame(ctx, new, ref, sn, sr, psf, psf, scal, 1320, 40)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for rep in range(n):
        outs = G.run_zogy_frame(ctx, new, ref, sn, sr, psf, psf, scal, 1320, 40)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print('nsub %2d: %.3f ms per call, %.1f us per sub-image' % (nsub, ms, 1e3 * ms / nsub), flush=True)
    del new, ref, sn, sr, outs
