"""CPU: FITS reader/writer round trips and the frame-farm partition (incl. a world_size-2
gloo run of the shard + gather logic)."""
import os
import subprocess
import sys

import numpy as np

from blackbox_amd import farm, fitsio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fits_roundtrip(tmp_path):
    rs = np.random.RandomState(0)
    for arr in (rs.randint(0, 65536, (37, 53)).astype(np.uint16), rs.normal(0, 1, (5, 7)).astype(np.float32),
                rs.randint(0, 255, (9, 4)).astype(np.uint8)):
        p = str(tmp_path / 'a.fits')
        hdr = {'EXPTIME': (60.0, '[s] exposure'), 'OBJECT': ('field 16123', 'name'), 'FLAG-P': (True, 'bool'),
               'NCOUNT': (12345, 'int'), 'DATE-OBS': ('2024-08-07T01:02:03.5', 'utc')}
        fitsio.write_image(p, arr, hdr)
        assert os.path.getsize(p) % 2880 == 0
        back, h = fitsio.read_image(p, get_header=True)
        assert back.dtype == arr.dtype and np.array_equal(back, arr)
        assert h['EXPTIME'][0] == 60.0 and h['OBJECT'][0] == 'field 16123' and h['FLAG-P'][0] is True
        assert h['NCOUNT'][0] == 12345 and h['DATE-OBS'][0] == '2024-08-07T01:02:03.5'
        assert fitsio.read_image(p, dtype='float32').dtype == np.float32


def test_shard_partition():
    files = ['f%03d' % i for i in range(23)]
    for world in (1, 2, 3, 8):
        parts = [farm.shard(files, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == files                      # disjoint and complete
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch.distributed as dist
from blackbox_amd import farm
dist.init_process_group('gloo')
files = ['frame%%02d.fits' %% i for i in range(11)]
mine = farm.shard(files)
done = [f.replace('.fits', '_red.fits') for f in mine]          # stand-in for the per-frame reduction
allr = farm.gather_results(done)
assert sorted(allr) == sorted(f.replace('.fits', '_red.fits') for f in files), allr
assert len(mine) in (5, 6)
dist.barrier()
if dist.get_rank() == 0:
    print('FARM_OK', len(allr))
dist.destroy_process_group()
'''


def test_farm_two_ranks_gloo(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                        '--master-addr', '127.0.0.1', '--master-port', '29533', str(script)],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'FARM_OK 11' in r.stdout


def test_eight_ranks_stay_inside_the_cpu_budget_of_a_node(monkeypatch):
    """configs[3] / [4]: eight ranks on one node, one GPU each (blackbox.py:363-379 farms on the host's cores).  The threads
    and worker processes a rank starts are sized from the cores the process may use (cgroup quota / affinity) divided by the
    ranks of the node: on a 64- and a 128-core node the eight ranks together ask for no more processes than there are
    cores, every rank keeps at least two fit workers, and a rank alone on a 16-core box gets the twelve that were measured
    best there."""
    from blackbox_amd import pipeline
    for cores, world in ((64, 8), (128, 8), (16, 1), (32, 8), (8, 2)):
        monkeypatch.setattr(pipeline, 'cpu_budget', lambda c=cores: c)
        monkeypatch.setenv('LOCAL_WORLD_SIZE', str(world))
        monkeypatch.delenv('BBX_HOST_WORKERS', raising=False)
        w = pipeline.default_workers()
        assert 2 <= w <= 12
        # per rank: the fit workers + the orchestrating thread + one lane thread that is busy at a time (the others sleep in
        # bbx_wait, measured 6-10 ms of CPU per frame for all of them: DESIGN section 1)
        assert world * (w + 2) <= max(cores, world * 4), (cores, world, w)
    monkeypatch.setattr(pipeline, 'cpu_budget', lambda: 16)
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '1')
    assert pipeline.default_workers() == 12
