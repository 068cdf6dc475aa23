"""GPU box: which Python lines of the host layer cause device copies / small torch kernels in one serial frame
(reduce_object + optimal_subtraction): counts per call site of Tensor.to / cpu / copy_ / item / tolist / torch.tensor /
as_tensor / zeros / full / empty-like constructors with a device, and of torch operators on device tensors."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
import bench
from blackbox_amd import reduce as R, zogy as G

counts = collections.Counter()
ON = [False]
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if fr.filename.startswith(ROOT) and 'copy_sites' not in fr.filename:
            return '%s:%d' % (os.path.relpath(fr.filename, ROOT), fr.lineno)
    return '?'


def wrap(obj, name, tag):
    f = getattr(obj, name)

    def g(*a, **k):
        if ON[0]:
            counts[(tag, site())] += 1
        return f(*a, **k)
    setattr(obj, name, g)


for n in ('to', 'cpu', 'copy_', 'item', 'tolist', 'contiguous', 'clone', 'zero_', 'fill_', '__getitem__', '__setitem__', 'float', 'double', 'long', 'int'):
    wrap(torch.Tensor, n, 'Tensor.' + n)
for n in ('tensor', 'as_tensor', 'zeros', 'ones', 'full', 'stack', 'cat', 'where', 'zeros_like', 'empty_like', 'arange', 'from_numpy'):
    wrap(torch, n, 'torch.' + n)

ctx = R.Context(0)
dev = ctx.device
ysz, xsz = 5280, 1320
raw, flat, bpm, ex = bench.synth_frame_device(torch, dev, ysz, xsz, 20, 180, 4000, 'u16', extras=True, ntrans=50)
ref, ref_mask = bench.synth_reference(torch, dev, ex.pop('scene0'), 4000)
rs = np.random.RandomState(0)
coeffs = np.zeros((16, 16)); coeffs[~np.eye(16, dtype=bool)] = rs.uniform(0, 2e-4, 240)
psf = torch.from_numpy(bench.moffat_stamp(25, 4.0)).to(dev)
for i in range(3):
    ON[0] = i == 2
    data, mask, h, hm = R.reduce_object(ctx, raw, {}, 'ML1', mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0)
    res = G.optimal_subtraction(ctx, data, ref, mask, ref_mask, psf, psf, fratio=1.0, dx=0.03, dy=0.03, cat_extract=True,
                                ref_is_bkgsub=True, ref_bkg_std_mini=np.full((176, 176), 8.0, np.float32))
    ctx.sync()
ON[0] = False
tot = collections.Counter()
for (tag, s), c in counts.items():
    tot[tag] += c
print('per frame by kind:', dict(tot))
for (tag, s), c in sorted(counts.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print('%4d  %-22s %s' % (c, tag, s))
