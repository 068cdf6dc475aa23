"""GPU box: time bbx_funpack_tiles on a full-size fpacked raw frame (uint16, 10600 x 12000) and check the pixels"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from blackbox_amd import reduce as R, fpack as P
ctx = R.Context(0)
g = torch.Generator(device=ctx.device); g.manual_seed(2)
raw = (2000 + 30 * torch.randn(10600, 12000, device=ctx.device, generator=g)).clamp(0, 65535).to(torch.int32)
raw[:, 5000:5003] = 65535; raw[100:110, :] = 0
raw = raw.to(torch.uint16) if hasattr(torch, 'uint16') else raw
path = P.fpack_image(ctx, '/dev/shm/fu_time.fits', raw, {'A': 1})
back, hdr = P.funpack_image(ctx, path)
assert bool((back.to(torch.int32) == raw.to(torch.int32)).all()), 'pixels differ'
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): back, hdr = P.funpack_image(ctx, path)
    e1.record(); torch.cuda.synchronize()
    print('funpack_image (file read + H2D + decode) %.2f ms per frame' % (e0.elapsed_time(e1) / 5))
os.unlink(path)
