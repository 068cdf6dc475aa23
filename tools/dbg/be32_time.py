import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from blackbox_amd import fitsio, reduce as R
ctx = R.Context(0)
a = np.random.RandomState(1).standard_normal((10560, 10560)).astype(np.float32)
p = '/dev/shm/t_be.fits'
fitsio.write_image(p, a)
for rep in range(3):
    t0 = time.time(); x = torch.from_numpy(np.ascontiguousarray(fitsio.read_image(p, dtype=np.float32))).to(ctx.device); torch.cuda.synchronize(); t1 = time.time()
    y = R.image_to_device(ctx, p, np.float32); torch.cuda.synchronize(); t2 = time.time()
    t3 = time.time(); h = fitsio.read_hdus(p); t4 = time.time()
    print('host swap path %.3f s   device swap path %.3f s   (read_hdus alone %.3f s)  equal %s' % (t1 - t0, t2 - t1, t4 - t3, bool((x == y).all())))
import os; os.unlink(p)
