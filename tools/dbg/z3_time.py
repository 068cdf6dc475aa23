"""time the kernels of bbx_zogy_frame alone on a full-size frame (the library's own per-kernel event slots), n reps;
BBX_LIB_PATH selects a scratch build (tools/exp/zvar.sh)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
import bench
from blackbox_amd import reduce as R, zogy as G, _lib
ctx = R.Context(0)
dev = ctx.device
ny = nx = 10560
g = torch.Generator(device=dev); g.manual_seed(1)
new = (20 * torch.randn(ny, nx, device=dev, generator=g)).contiguous()
ref = (8 * torch.randn(ny, nx, device=dev, generator=g)).contiguous()
sn = torch.full((ny, nx), 20.0, device=dev); sr = torch.full((ny, nx), 8.0, device=dev)
S = int(os.environ.get('S', 49))
psf = torch.from_numpy(np.repeat(bench.moffat_stamp(S, 4.0)[None], 64, 0)).to(dev)
scal = np.tile(np.array([[20, 8, 1, 1, 0.03, 0.03]], np.float32), (64, 1))
if os.environ.get('MINI'):
    # the sigma maps as mini images (bbx_zogy_frame_mini): new per channel, reference one patch
    rs = np.random.RandomState(3)
    sn = G.MiniImage(ctx, (20 + rs.random_sample((176, 176))).astype(np.float32), 60, interp_Xchan=False)
    sr = G.MiniImage(ctx, (8 + rs.random_sample((176, 176))).astype(np.float32), 60, interp_Xchan=True)
for rep in range(2):
    outs = G.run_zogy_frame(ctx, new, ref, sn, sr, psf, psf, scal, 1320, 40)
torch.cuda.synchronize()
_lib.lib.bbx_profile_enable(ctx.h, 1)
n = int(os.environ.get('N', 6))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for rep in range(n):
    outs = G.run_zogy_frame(ctx, new, ref, sn, sr, psf, psf, scal, 1320, 40)
e1.record(); torch.cuda.synchronize()
ms = (C.c_double * 14)(); calls = (C.c_int32 * 14)()
_lib.lib.bbx_profile_read(ctx.h, ms, calls, 14)
names = {7: 'final_rows', 8: 'psf_cols', 9: 'psf_rows', 10: 'img_rows(x2)', 11: 'img_cols', 12: 'var_cols', 13: 'psf_rowdft'}
per = {names[k]: ms[k] / n for k in names if calls[k]}
print(os.environ.get('BBX_LIB_PATH', 'product') + (' MINI' if os.environ.get('MINI') else ''), 'total %.3f ms |' % (e0.elapsed_time(e1) / n), ' '.join('%s %.3f' % kv for kv in per.items()),
      '| Scorr std %.4f' % float(outs[2].std()))
