"""GPU box: time the spline zoom variants (write-only sigma image, fused subtraction) with events; BBX_LIB_PATH picks a scratch build"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from blackbox_amd import reduce as R, zogy as G
ctx = R.Context(0)
dev = ctx.device
ny = nx = 10560
data = torch.randn(ny, nx, device=dev)
work = torch.empty_like(data)
mini = torch.from_numpy(np.random.RandomState(0).normal(100, 5, (176, 176)).astype(np.float32)).to(dev)
def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(os.environ.get('BBX_LIB_PATH', 'product'),
      'std zoom (write only, per channel) %.3f ms' % t(lambda: G.mini2back(ctx, mini, (ny, nx), interp_Xchan=False)),
      '| bkg zoom + subtract into %.3f ms' % t(lambda: G.mini2back(ctx, mini, (ny, nx), subtract_from=data, subtract_into=work)))
print('prefilter alone: full %.3f ms, per channel %.3f ms' % (t(lambda: G.device_zoom_coefficients(ctx, mini, None)), t(lambda: G.device_zoom_coefficients(ctx, mini, (88, 22)))))
bk = torch.empty_like(data)
print('in-place subtract %.3f ms' % t(lambda: G.mini2back(ctx, mini, (ny, nx), subtract_from=data, want_bkg=False)))
