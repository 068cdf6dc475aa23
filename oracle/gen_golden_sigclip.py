"""Golden vectors for sigma_clipped_stats(x, mask_value=0) from the reference environment's
astropy (run with /opt/conda/bin/python3.9; astropy 4.3.1):  inputs are regenerated from the
seeds, outputs are stored in tests/golden/sigclip.npz."""
import json
import os

import numpy as np

# astropy 4.3 still looks these removed numpy names up at import time (same shim as _refload.py)
for _n, _f in {'asscalar': lambda a: a.item(), 'alen': len}.items():
    if not hasattr(np, _n):
        setattr(np, _n, _f)
from astropy.stats import sigma_clipped_stats
import astropy

out = {}
meta = []
for seed, n, frac_out in ((1, 5000, 0.02), (2, 20001, 0.05), (3, 777, 0.0)):
    rs = np.random.RandomState(seed)
    x = rs.normal(10, 3, n).astype(np.float32)
    k = int(frac_out * n)
    if k:
        x[rs.randint(0, n, k)] += 100
    x[rs.randint(0, n, 20)] = 0
    mean, med, std = sigma_clipped_stats(x, mask_value=0)
    out['res_%d' % seed] = np.array([mean, med, std], np.float64)
    meta.append(dict(seed=seed, n=n, frac_out=frac_out))
out['meta'] = json.dumps(dict(cases=meta, astropy=astropy.__version__, numpy=np.__version__))
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'sigclip.npz'), **out)
print(out)
