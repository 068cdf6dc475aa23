"""GPU box: [n] serial frames of the headline workload (reduce + optimal_subtraction) on one
stream, for `rocprofv3 --kernel-trace --stats` (kernels alone on the GPU):
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps -o r -- python3 tools/prof_serial.py 5
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    import torch
    import bench
    from blackbox_amd import reduce as R, zogy as G
    ctx = R.Context(0)
    dev = ctx.device
    ysz, xsz = 5280, 1320
    raw, flat, bpm, ex = bench.synth_frame_device(torch, dev, ysz, xsz, 20, 180, 4000, 'u16', extras=True, ntrans=50)
    ref, ref_mask = bench.synth_reference(torch, dev, ex.pop('scene0'), 4000)
    rs = np.random.RandomState(0)
    coeffs = np.zeros((16, 16)); coeffs[~np.eye(16, dtype=bool)] = rs.uniform(0, 2e-4, 240)
    zi = bench.zogy_inputs(torch, dev, 8, 8, 49, 60, 2 * ysz, 8 * xsz)          # the SURVEY 8d ZOGY inputs of the headline workload
    for i in range(n + 1):
        data, mask, h, hm = R.reduce_object(ctx, raw, {}, 'ML1', mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0)
        res = G.optimal_subtraction(ctx, data, ref, mask, ref_mask, cat_extract=True, **zi)
        ctx.sync()
        del res
    print('done', n + 1, 'frames')


if __name__ == '__main__':
    main()
