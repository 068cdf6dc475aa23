"""debug: determinism of optimal_subtraction + where the full-size float32 errors sit"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'oracle'))
import numpy as np, torch
import bench, zogy_core as Z
from blackbox_amd import reduce as R, zogy as G, synth
import bbx_oracle as O

ctx = R.Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)
# --- small determinism check
YS, XS = 120, 330
case = synth.make_case(YS, XS, 77, tel='ML1', os_y=20, os_x=45, n_stars=60, n_sat=2, n_cr=40)
d0, m0, h0, _ = R.reduce_object(ctx, dev(case['raw']), {}, 'ML1', mflat=dev(case['flat']), bpm=dev(case['bpm']),
                                xtalk_coeffs=O.xtalk_coeffs(case['xtalk']), exptime=60.0, ysize_chan=YS, xsize_chan=XS)
rs = np.random.RandomState(3)
ref = (d0.cpu().numpy() - 100.0 + rs.normal(0, 4, d0.shape)).astype(np.float32)
psf = dev(bench.moffat_stamp(15, 3.5))
outs = []
for k in range(3):
    res = G.optimal_subtraction(ctx, d0, dev(ref), m0, torch.zeros_like(m0), psf, psf, subimage_size=120, subimage_border=10,
                                bkg_boxsize=20, cat_extract=(k != 1))
    ctx.sync()
    outs.append({key: res[key].cpu().numpy().copy() for key in ('D', 'Scorr', 'Fpsf', 'Fpsferr')})
for key in outs[0]:
    print('determinism', key, [float(np.abs(outs[0][key] - o[key]).max()) for o in outs[1:]], 'nan', int(np.isnan(outs[0][key]).sum()))
bad = np.argwhere(outs[0]['Scorr'] != outs[1]['Scorr'])
print('n differing Scorr px', len(bad), bad[:5])

# --- full-size error map
YSZ, XSZ = 5280, 1320
raw, flat, bpm, ex = bench.synth_frame_device(torch, ctx.device, YSZ, XSZ, 20, 180, 3000, 'u16', extras=True, ntrans=40)
data, mask, header, hm = R.reduce_object(ctx, raw, {}, 'ML1', mflat=flat, bpm=bpm, exptime=60.0)
refi, refm = bench.synth_reference(torch, ctx.device, ex['scene0'], 3000)
pn, pr = bench.moffat_stamp(25, 4.0), bench.moffat_stamp(25, 3.6)
res = G.optimal_subtraction(ctx, data, refi, mask, refm, dev(pn), dev(pr), fratio=1.0, dx=0.03, dy=0.02)
ctx.sync()
SIZE, BORDER = 1320, 40; L = 1400; NY = NX = 10560
def embed(p, dt):
    k = np.zeros((L, L), dt); h = p.shape[0] // 2
    for j in range(p.shape[0]):
        for i in range(p.shape[1]):
            k[(j - h) % L, (i - h) % L] = p[j, i]
    return k
def cut(t, sy, sx):
    out = np.zeros((L, L), np.float32)
    y0, x0 = sy * SIZE - BORDER, sx * SIZE - BORDER
    ya, yb, xa, xb = max(y0, 0), min(y0 + L, NY), max(x0, 0), min(x0 + L, NX)
    out[ya - y0:yb - y0, xa - x0:xb - x0] = t[ya:yb, xa:xb].cpu().numpy()
    return out
sy, sx = 3, 4
k = sy * 8 + sx
N, Rr = cut(res['data_bkgsub'], sy, sx), cut(res['ref_bkgsub'], sy, sx)
Vn, Vr = cut(res['var_new'], sy, sx), cut(res['var_ref'], sy, sx)
sn, sr = res['scal'][k, 0], res['scal'][k, 1]
D32 = Z.run_zogy(N, Rr, embed(pn, np.float32), embed(pr, np.float32), sn, sr, 1.0, 1.0, Vn, Vr, 0.03, 0.02)[0]
# float64 evaluation of D
f2, if2 = np.fft.fft2, np.fft.ifft2
Nh, Rh, Pnh, Prh = f2(N.astype(np.float64)), f2(Rr.astype(np.float64)), f2(embed(pn, np.float64)), f2(embed(pr, np.float64))
sn, sr = float(sn), float(sr)
den = sn * sn * np.abs(Prh) ** 2 + sr * sr * np.abs(Pnh) ** 2
fD = 1.0 / np.sqrt(sn * sn + sr * sr)
D64 = (if2((Prh * Nh - Pnh * Rh) / np.sqrt(den)).real / fD)
inner = (slice(BORDER, BORDER + SIZE), slice(BORDER, BORDER + SIZE))
got = res['D'][sy * SIZE:(sy + 1) * SIZE, sx * SIZE:(sx + 1) * SIZE].cpu().numpy()
for name, a in (('hip', got), ('np32', D32[inner])):
    e = np.abs(a - D64[inner])
    j, i = np.unravel_index(np.argmax(e), e.shape)
    print(name, 'vs float64: max err', e.max(), 'at', (j, i), 'rms', np.sqrt((e ** 2).mean()), 'p99.9', np.percentile(e, 99.9))
    print('   N around', N[inner][max(j-2,0):j+3, max(i-2,0):i+3].max(), 'row max', np.abs(N[inner][j]).max(), 'col max', np.abs(N[inner][:, i]).max(),
          'sub max', np.abs(N).max(), 'l2', np.sqrt((N.astype(np.float64) ** 2).sum()))
e = np.abs(got - D32[inner])
print('hip vs np32 max', e.max(), 'rms', np.sqrt((e**2).mean()))
rowmax = np.abs(N[inner]).max(axis=1); colmax = np.abs(N[inner]).max(axis=0)
erow = np.abs(got - D64[inner]).max(axis=1); ecol = np.abs(got - D64[inner]).max(axis=0)
top = np.argsort(erow)[-5:]
print('rows with largest err', [(int(r), float(erow[r]), float(rowmax[r])) for r in top])
top = np.argsort(ecol)[-5:]
print('cols with largest err', [(int(r), float(ecol[r]), float(colmax[r])) for r in top])
