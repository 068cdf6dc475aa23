"""The generated register DFTs (blackbox_amd/csrc/bbx_fft_gen.h, tools/gen_fft.py): the committed header is what the
generator writes, and every size agrees with numpy.fft in both directions (host build of the same source with the
portable definitions of the pair primitives; the device build spells them as packed instructions)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = '/opt/rocm/lib/llvm/bin/clang++'


def test_committed_header_is_the_generators_output():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gen_fft
    assert open(os.path.join(ROOT, 'blackbox_amd', 'csrc', 'bbx_fft_gen.h')).read() == gen_fft.gen(gen_fft.SIZES)


@pytest.mark.skipif(not os.path.exists(CLANG) and shutil.which('clang++') is None, reason='no clang++ for the host build')
def test_register_dfts_against_numpy_fft():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_fft.py'), '--check'], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0, out.stdout.decode()
    assert b'agree with numpy.fft' in out.stdout
