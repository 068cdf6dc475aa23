import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    # the ABI / host tests load the in-tree libraries: build them once if a fresh checkout has
    # none yet (hipcc cross-compiles gfx950 without a GPU); a GPU box gets them with the snapshot
    # ... and rebuild when a source is newer than the library (make decides; needs hipcc)
    lib = os.path.join(ROOT, 'blackbox_amd', 'libbbx_hip.so')
    import glob
    import shutil
    srcs = glob.glob(os.path.join(ROOT, 'blackbox_amd', 'csrc', '*')) + [os.path.join(ROOT, 'include', 'bbx.h')]
    srcs = [f for f in srcs if f.endswith(('.hip', '.h'))]
    # (not on a gpurun box -- GRAFT_REPO_ROOT is set there: the snapshot's file times say nothing)
    stale = (os.path.isfile(lib) and 'GRAFT_REPO_ROOT' not in os.environ
             and any(os.path.getmtime(f) > os.path.getmtime(lib) for f in srcs))
    hipcc = shutil.which('hipcc') or (os.path.isfile('/opt/rocm/bin/hipcc') and '/opt/rocm/bin/hipcc')
    if (not os.path.isfile(lib) or (stale and hipcc)) and os.path.isfile(os.path.join(ROOT, 'Makefile')):
        import subprocess
        subprocess.run(['make', '-C', ROOT, '-j8', 'all'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
