"""full-size frame with a trail: HIP detector info vs oracle info"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'oracle'))
import numpy as np, torch
import bench
import sattrail as S
from blackbox_amd import reduce as R

if __name__ == '__main__':
    ctx = R.Context(0)
    YSZ, XSZ = 5280, 1320
    TRAIL = (0.0, 2100.0, float(8 * XSZ), 6400.0, float(sys.argv[1]) if len(sys.argv) > 1 else 90.0, 6.0)
    raw, flat, bpm, ex = bench.synth_frame_device(torch, ctx.device, YSZ, XSZ, 20, 180, 3000, 'u16', extras=True, ntrans=40, trail=TRAIL)
    stages = {}
    data, mask, header, hm = R.reduce_object(ctx, raw, {}, 'ML1', mflat=flat, bpm=bpm, exptime=60.0, stages=stages)
    pre = stages['data_xtalk'] if 'data_xtalk' in stages else data
    m0 = (mask & ~16)
    d_mask = m0.clone()
    d_n, d_info = R.sat_detect(ctx, pre, {}, d_mask, {})
    ctx.sync()
    print('HIP info', d_info.cpu().numpy(), int(d_n.item()), flush=True)
    t = time.time()
    b = S.bin2(pre.cpu().numpy())
    kept, edge, img, p1, p2 = S.edges(b, True)
    print('oracle p1 p2', p1, p2, 'edge', edge.sum(), 'kept', kept.sum(), time.time() - t, flush=True)
    acc, off = S.hough(kept)
    flat_ = int(np.argmax(acc.T)); k, r = divmod(flat_, acc.shape[0])
    print('oracle best votes', acc[r, k], 'k', k, 'theta', S.THETA_DEG[k], 'rho', r - off, time.time() - t, flush=True)
    # the true line in binned coordinates
    xa, ya, xb, yb = [v / 2 for v in TRAIL[:4]]
    ang = np.degrees(np.arctan2(yb - ya, xb - xa)) + 90
    print('true normal angle', ang % 180, flush=True)
    m_o, n_o, info_o = S.detect(pre.cpu().numpy())
    print('oracle detect', n_o, {k: v for k, v in info_o.items()}, flush=True)
    lab_sizes = np.bincount(__import__('scipy.ndimage', fromlist=['label']).label(edge, structure=np.ones((3, 3)))[0].ravel())[1:]
    print('edge components', lab_sizes.size, 'sizes >= 60:', (lab_sizes >= 60).sum(), 'largest', np.sort(lab_sizes)[-10:], flush=True)
