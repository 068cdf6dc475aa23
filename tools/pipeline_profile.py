"""cProfile of the orchestrating thread of FramePipeline (GPU box)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

if __name__ == '__main__':
    import torch
    import bench
    from blackbox_amd import reduce as R
    from blackbox_amd.pipeline import FramePipeline, HostPool
    ctx = R.Context(0)
    ysz, xsz, os_y, os_x = 5280, 1320, 20, 180
    raw, flat, bpm = bench.synth_frame_device(torch, ctx.device, ysz, xsz, os_y, os_x, 2000, 'u16')
    geom = R.geometry(raw.shape, ysz, xsz)
    pool = HostPool(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
    pipe = FramePipeline(ctx, 'ML1', geom, mflat=flat, bpm=bpm, pool=pool, depth=int(sys.argv[2]) if len(sys.argv) > 2 else 12, lanes=int(sys.argv[3]) if len(sys.argv) > 3 else 2)
    pipe.run([(raw, {}) for _ in range(6)])
    pipe.t_stats = [0.0, 0.0, 0.0, 0]
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    pipe.run([(raw, {}) for _ in range(120)])
    pr.disable()
    dt = time.perf_counter() - t0
    print('fps', 120 / dt, 'stats ms/frame', [1e3 * t / 120 for t in pipe.t_stats[:3]])
    pstats.Stats(pr).sort_stats('tottime').print_stats(22)
    pipe.close(); pool.close()
