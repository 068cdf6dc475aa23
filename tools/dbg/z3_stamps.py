"""phase durations inside the kernels of bbx_zogy_frame from a -DZ3_STAMPS scratch build (tools/exp/zvar.sh): thread 0 of every
workgroup stamps the shader clock at its phase boundaries; prints the median / mean length of every phase per kernel, the
workgroup's whole life and how many workgroups overlap in time.  BBX_LIB_PATH must name the stamped build."""
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
S = int(os.environ.get('S', 49))
g = torch.Generator(device=dev); g.manual_seed(1)
new = (20 * torch.randn(ny, nx, device=dev, generator=g)).contiguous()
ref = (8 * torch.randn(ny, nx, device=dev, generator=g)).contiguous()
sn = torch.full((ny, nx), 20.0, device=dev); sr = torch.full((ny, nx), 8.0, device=dev)
psf = torch.from_numpy(np.repeat(bench.moffat_stamp(S, 4.0)[None], 64, 0)).to(dev)
scal = np.tile(np.array([[20, 8, 1, 1, 0.03, 0.03]], np.float32), (64, 1))
for rep in range(2):
    outs = G.run_zogy_frame(ctx, new, ref, sn, sr, psf, psf, scal, 1320, 40)
torch.cuda.synchronize()
NWG = 24576
buf = torch.zeros(6 * NWG * 16, dtype=torch.int64, device=dev)
lib = _lib.lib
lib.bbx_z3_stamps.argtypes = [C.c_void_p]; lib.bbx_z3_stamps.restype = C.c_int
assert lib.bbx_z3_stamps(C.c_void_p(buf.data_ptr())) == 0
outs = G.run_zogy_frame(ctx, new, ref, sn, sr, psf, psf, scal, 1320, 40)
torch.cuda.synchronize()
lib.bbx_z3_stamps(C.c_void_p(0))
a = buf.cpu().numpy().reshape(6, NWG, 16)
names = ['k_psf_cols', 'k_psf_rows', 'k_img_rows_both', 'k_img_cols', 'k_var_cols', 'k_final_rows']
phases = {
    'k_psf_cols': ['aux+zero', 'stampDFT_n', 'fft_n', 'park+zero', 'stampDFT_r', 'fft_r', 'coef loop(write A,B,Sd)', 'inv kr', 'store_win+unpark', 'inv kn', 'store_win+reduce'],
    'k_psf_rows': ['aux', 'load_u_pair', 'inv', 'square', 'fwd', 'store_t'],
    'k_img_rows_both': ['aux+load4frames+lds', 'fwd(N,R)', 'store_t', 'unpark V', 'fwd(V)', 'store_t'],
    'k_img_cols': ['aux+load TN(+fetch TR)', 'fwd N', 'park+pack TR', 'fwd R', 'coef loop(read A,B,Sd)', 'inv Sr', 'store+unpark', 'inv Sn', 'store+unpark', 'inv D', 'store'],
    'k_var_cols': ['aux+load k2n win', 'fwd k2n', 'coef+load TVn', 'fwd Vn', 'mul+load k2r win', 'fwd k2r', 'coef+load TVr', 'fwd Vr', 'combine', 'inv', 'store'],
    'k_final_rows': ['aux + first loads issued', 'the chunk of row blocks'],
}
for k, name in enumerate(names):
    st = a[k]
    used = st[:, 0] != 0
    st = st[used]
    if not len(st):
        continue
    nph = len(phases[name])
    if name == 'k_final_rows':
        st = st.copy(); st[:, 2] = st[:, 8]
    t = st[:, :nph + 1].astype(np.float64)
    d = np.diff(t, axis=1)
    life = t[:, nph] - t[:, 0]
    rt = st[:, 14].astype(np.float64)                       # 100 MHz
    span_us = (rt.max() - rt.min()) / 100.0
    clk = np.median(life) / 1.0
    print('%s: %d workgroups, kernel span (first start -> last start) %.0f us, workgroup life median %.0f cycles (mean %.0f, p90 %.0f)'
          % (name, len(st), span_us, np.median(life), life.mean(), np.percentile(life, 90)))
    for i, ph in enumerate(phases[name]):
        print('    %-28s median %7.0f  mean %7.0f  p90 %7.0f   %4.1f %%' % (ph, np.median(d[:, i]), d[:, i].mean(), np.percentile(d[:, i], 90),
                                                                        100 * d[:, i].mean() / life.mean()))
    xcc = st[:, 15] & 15
    print('    xcc histogram', np.bincount(xcc.astype(int), minlength=8).tolist())
    if name == 'k_var_cols' and st[:, 12].any():
        print('    FFT2X: first fwd k2n %.0f, repeated at once %.0f, again %.0f' % (np.median(st[:, 2] - st[:, 1].astype(np.float64)),
              np.median(st[:, 12].astype(np.float64) - st[:, 2]), np.median(st[:, 13].astype(np.float64) - st[:, 12])))
