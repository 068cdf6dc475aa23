import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'oracle'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import numpy as np, torch
os.environ.setdefault('PYTEST_CURRENT_TEST', 'x')
import importlib.util
spec = importlib.util.spec_from_file_location('tz', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'test_gpu_zogy_frame.py'))
T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)
from blackbox_amd import reduce as R, zogy as G
ctx = R.Context(0)
for (size, border, nsy, nsx, S) in [(48, 8, 2, 3, 11), (120, 10, 2, 4, 15)]:
    new, ref, sig_n, sig_r, pn, pr, scal = T.make(size, border, nsy, nsx, S, seed=size + 3 * border)
    for variant in ('full', 'dx0'):
        sc = scal.copy()
        if variant == 'dx0':
            sc[:, 4:] = 0
        want = T.oracle(new, ref, sig_n, sig_r, pn, pr, sc, size, border)
        got = G.run_zogy_frame(ctx, T.dev(ctx, new), T.dev(ctx, ref), T.dev(ctx, sig_n), T.dev(ctx, sig_r), T.dev(ctx, pn), T.dev(ctx, pr), sc, size, border, want_S=True)
        ctx.sync()
        for name, g, w in zip(('D', 'S', 'Scorr', 'Fpsf', 'Fpsferr'), got, want):
            g = g.cpu().numpy()
            e = np.abs(g - w)
            j, i = np.unravel_index(np.nanargmax(e), e.shape)
            print(size, border, variant, name, 'max err', np.nanmax(e), 'scale', np.nanmax(np.abs(w)), 'at', (j, i), 'rel med', np.nanmedian(e / np.maximum(np.abs(w), 1e-6)))
        vs_g = (got[4].cpu().numpy()) ** 2; vs_w = want[4] ** 2
        r = vs_g / vs_w
        print('   VS ratio min/med/max', np.nanmin(r), np.nanmedian(r), np.nanmax(r))
        # per sub-image median ratio
        for sy in range(nsy):
            print('   ', [float('%.5f' % np.nanmedian(r[sy*size:(sy+1)*size, sx*size:(sx+1)*size])) for sx in range(nsx)])
