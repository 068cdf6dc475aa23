"""time bbx_sat_trails alone on a full-size frame, n reps"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from blackbox_amd import reduce as R
ctx = R.Context(0)
dev = ctx.device
ny = nx = 10560
g = torch.Generator(device=dev); g.manual_seed(1)
data = (250 + 17 * torch.randn(ny, nx, device=dev, generator=g)).contiguous()
yy = torch.arange(ny, device=dev, dtype=torch.float32)[:, None]; xx = torch.arange(nx, device=dev, dtype=torch.float32)[None, :]
d = (xx * 0.4 + yy * 0.9165 - 6000.0)
data += 90 * torch.exp(-0.5 * (d / 2.5) ** 2)
mask = torch.zeros((ny, nx), dtype=torch.uint8, device=dev)
for rep in range(2):
    R.sat_detect(ctx, data, {}, mask, {})
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 5
for rep in range(n):
    dn, info = R.sat_detect(ctx, data, {}, mask, {})
e1.record(); torch.cuda.synchronize()
print('bbx_sat_trails ms', e0.elapsed_time(e1) / n, 'nsats', int(dn.item()), info.cpu().numpy())
