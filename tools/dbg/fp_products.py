"""GPU box: bbx_fpack_tiles on the images a frame of the headline workload leaves (reduced image, D, Scorr, Fpsf, the
detection limit = T-NSIGMA x Fpsferr): time per image next to what makes an image expensive -- rows that did not fit the
short stream buffer (second launch), bits per pixel, share of exactly equal neighbours."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import ctypes as C
import numpy as np, torch
import bench
from blackbox_amd import reduce as R, zogy as G, fpack as P, _lib

ctx = R.Context(0)
dev = ctx.device
ysz, xsz = 5280, 1320
variant = int(os.environ.get('VARIANT', 0))
raw, flat, bpm, ex = bench.synth_frame_device(torch, dev, ysz, xsz, 20 + variant, 180, 4000, 'u16', extras=True, ntrans=50)
ref, ref_mask = bench.synth_reference(torch, dev, ex.pop('scene0'), 4000)
rs = np.random.RandomState(0)
coeffs = np.zeros((16, 16)); coeffs[~np.eye(16, dtype=bool)] = rs.uniform(0, 2e-4, 240)
zi = bench.zogy_inputs(torch, dev, 8, 8, 49, 60, 2 * ysz, 8 * xsz)
data, mask, h, hm = R.reduce_object(ctx, raw, {}, 'ML1', mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0)
res = G.optimal_subtraction(ctx, data, ref, mask, ref_mask, cat_extract=True, **zi)
ctx.sync()
imgs = [('red', data, 16), ('D', res['D'], 16), ('Scorr', res['Scorr'], 16), ('Fpsf', res['Fpsf'], 16), ('limmag', (res['Fpsferr'] * 6.0).contiguous(), 16)]
if os.environ.get('SAVE'):
    for name, a, q in imgs:
        np.save(os.path.join(os.environ['SAVE'], 'fp_%s_rows.npy' % name), a[::500].cpu().numpy())
ny, nx = data.shape
stride = _lib.lib.bbx_fpack_tile_stride(nx, 4)
scratch = torch.empty(ny * stride, dtype=torch.uint8, device=dev)
tiles = torch.empty(ny * 24, dtype=torch.uint8, device=dev)
rnd = P._rnd(dev)


def run(a, q):
    _lib.check(_lib.lib.bbx_fpack_tiles(ctx.h, ny, nx, C.c_void_p(a.data_ptr()), -32, float(q), 1, C.c_void_p(rnd.data_ptr()),
                                        C.c_void_p(scratch.data_ptr()), C.c_void_p(tiles.data_ptr()), ctx.stream()), 'tiles')


def t(f, n=6):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, a, q in imgs:
    a = a.contiguous()
    ms = t(lambda: run(a, q))
    _lib.check(_lib.lib.bbx_set_option(ctx.h, 6, 1), 'opt')
    ms_h = t(lambda: run(a, q))
    _lib.check(_lib.lib.bbx_set_option(ctx.h, 6, 0), 'opt')
    run(a, q); torch.cuda.synchronize()
    tl = tiles.cpu().numpy().view(P._TILE_DT)
    eq = float((a[:, 1:] == a[:, :-1]).float().mean())
    zero = float((a == 0).float().mean())
    print('%-7s %.3f ms (plain paths %.3f)  %.2f bits/pixel  rows > 16 bits/pixel: %d  refused %d  equal neighbours %.4f  zeros %.4f  zscale median %.3g  min %.3g max %.3g' % (
        name, ms, ms_h, 8.0 * tl['nbytes'].sum() / a.numel(), int((tl['nbytes'] > 2 * nx).sum()), int((tl['flag'] != 0).sum()), eq, zero,
        float(np.median(tl['zscale'])), float(a.min()), float(a.max())))
