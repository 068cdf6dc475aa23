import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
import bench
from blackbox_amd import reduce as R, _lib
ctx = R.Context(0)
raw, flat, bpm = bench.synth_frame_device(torch, ctx.device, 5280, 1320, 20, 180, 2000, 'u16')
geom = R.geometry(raw.shape, 5280, 1320)
h, hm = {}, {}
R.gain_corr(h, 'ML1')
sol = R.os_solve(ctx, raw, h, 'ML1', geom)
data, mask = R.calibrate(ctx, raw, sol, h, hm, 'ML1', geom, mflat=flat, bpm=bpm)
R.edge_fill(ctx, data, mask, geom); R.edge_fill(ctx, data, mask, geom)
ctx.sync()
lib = _lib.lib
lib.bbx_debug_ws_ptr.restype = C.c_int
ptr, nb = C.c_void_p(), C.c_size_t()
WS_SEL = 12
lib.bbx_debug_ws_ptr(ctx.h, WS_SEL, C.byref(ptr), C.byref(nb))
o_seg = 64 * 64 * 64
buf = torch.empty(64 * 64, dtype=torch.uint8, device=ctx.device)
lib.bbx_copy_async(C.c_void_p(buf.data_ptr()), C.c_void_p(ptr.value + o_seg), 64 * 64, 2, ctx.stream())
ctx.sync()
b = buf.cpu().numpy()
for sg in range(16):
    rec = b[sg * 64:(sg + 1) * 64]
    lo, hi = np.frombuffer(rec[0:8].tobytes(), np.float32)
    nsample, nbuf = np.frombuffer(rec[8:16].tobytes(), np.uint32)
    below, n = np.frombuffer(rec[16:32].tobytes(), np.uint64)
    fail = np.frombuffer(rec[32:36].tobytes(), np.uint32)[0]
    res = np.frombuffer(rec[40:48].tobytes(), np.float32)
    ch = data[(sg // 8) * 5280:(sg // 8 + 1) * 5280, (sg % 8) * 1320:(sg % 8 + 1) * 1320]
    print(sg, 'lo/hi', lo, hi, 'nsample', nsample, 'nbuf', nbuf, 'below', below, 'n', n, 'fail', fail, 'res', res, 'frac', nbuf / max(1, n))
af = torch.empty(4, dtype=torch.int32, device=ctx.device)
lib.bbx_copy_async(C.c_void_p(af.data_ptr()), C.c_void_p(ptr.value + 268288), 16, 2, ctx.stream())
ctx.sync()
print('anyfail words', af.cpu().numpy())
