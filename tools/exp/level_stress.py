"""GPU box: every frame needs LA-Cosmic's background level while six lanes keep the GPU busy --
the cooperative on-demand select (k_lac_bg_frame) under load, feed kept off."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from blackbox_amd import reduce as R
from blackbox_amd.pipeline import FramePipeline, HostPool


class NoFeed(FramePipeline):
    def _finalize(self, f):
        super()._finalize(f)
        self.needed = getattr(self, 'needed', 0) + int(f.h_out[2].numpy()[15])
        self.level_feed_left = 0


if __name__ == '__main__':
    ctx = R.Context(0)
    ysz, xsz = 5280, 1320
    raw, flat, bpm = bench.synth_frame_device(torch, ctx.device, ysz, xsz, 20, 180, 2000, 'u16')
    geom = R.geometry(raw.shape, ysz, xsz)
    for (j, i) in ((3000, 4000), (7001, 123)):
        bpm[j - 2:j + 3, i - 2:i + 3] |= 1
        bpm[j, i] = 0
        iy, ix = j // ysz, i // xsz
        raw[iy * (ysz + 20) + (j - iy * ysz) + (0 if iy == 0 else 20), ix * (xsz + 180) + (i - ix * xsz)] = 20000
    pool = HostPool()
    pipe = NoFeed(ctx, 'ML1', geom, mflat=flat, bpm=bpm, pool=pool, depth=18, lanes=6)
    pipe.run([(raw, {}) for _ in range(12)])
    pipe.needed = 0
    t0 = time.perf_counter()
    n = pipe.run([(raw, {}) for _ in range(240)])
    dt = time.perf_counter() - t0
    print('frames', n, 'needed the level', pipe.needed, 'frames/s', round(n / dt, 1))
    pipe.close(); pool.close()
