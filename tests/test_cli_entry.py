"""CPU: the operator-level entry points of blackbox.py (the reference's blackbox.py:363-379, 933-999) -- no GPU needed:
WrapException carries the formatted traceback through pickling, configure() hands the settings to pool workers
through the environment, and without a GPU the product fails loudly instead of falling back to anything."""
import importlib.util
import os
import pickle

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


def load_cli():
    spec = importlib.util.spec_from_file_location('bbx_cli_entry', os.path.join(ROOT, 'blackbox.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_wrapexception_formats_and_pickles():
    cli = load_cli()
    try:
        try:
            raise KeyError('DATE-OBS')
        except KeyError:
            raise cli.WrapException()
    except cli.WrapException as e:
        assert 'KeyError' in e.formatted and 'Traceback' in e.formatted and 'Original traceback' in str(e)
        import sys
        sys.modules['bbx_cli_entry'] = cli                     # so that pickle finds the class by module name
        e2 = pickle.loads(pickle.dumps(e))
        assert isinstance(e2, cli.WrapException) and e2.formatted == e.formatted


def test_configure_and_loud_failure_without_gpu(monkeypatch):
    cli = load_cli()
    monkeypatch.delenv(cli._ENV_KEY, raising=False)
    with pytest.raises(cli.WrapException) as ei:
        cli.try_blackbox_reduce('ML1_nothing.fits')            # not configured
    assert 'configure' in ei.value.formatted
    cli.configure(['--telescope', 'BG3', '--red_dir', '/tmp/x'])
    import json
    assert json.loads(os.environ[cli._ENV_KEY]) == ['--telescope', 'BG3', '--red_dir', '/tmp/x']
    assert cli.telescope_of(cli.build_parser().parse_args(['--telescope', 'BG3']), '/data/BG4_20240101_raw.fits') == 'BG4'
    assert cli.pool_func(len, ['a', 'bb'], nproc=1) == [1, 2]
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(cli.WrapException) as ei:
            cli.try_blackbox_reduce('ML1_nothing.fits')
        assert 'needs a GPU' in ei.value.formatted or 'libbbx_hip' in ei.value.formatted
    monkeypatch.delenv(cli._ENV_KEY, raising=False)


def test_list_split_over_processes(monkeypatch, tmp_path, capsys):
    """--image_list over several pipelined processes on one GPU (list_processes / reduce_list_in_processes): how many, which
    files and cores each child gets, the flags it is started with, results back in the order of the list, None for the
    files of a child that died.  (The children are stand-ins here: no GPU.)"""
    import json
    import subprocess
    cli = load_cli()
    ap = cli.build_parser()
    # how many processes: by the cores of this process, only for a list that is long enough, never on top of --nproc
    for budget, world, nfiles, extra, want in ((16, 1, 96, [], 2), (16, 1, 8, [], 1), (8, 1, 96, [], 1), (64, 8, 96, [], 1), (128, 8, 96, [], 2),
                                               (16, 1, 96, ['--list_procs', '1'], 1), (4, 1, 96, ['--list_procs', '3'], 3),
                                               (16, 1, 96, ['--nproc', '4'], 1)):
        monkeypatch.setenv('BBX_CPU_BUDGET', str(budget))
        monkeypatch.setenv('LOCAL_WORLD_SIZE', str(world))
        monkeypatch.delenv('BBX_LIST_PROCS', raising=False)
        args = ap.parse_args(['--image_list', 'l.txt'] + extra)
        assert cli.list_processes(args, nfiles) == want, (budget, world, nfiles, extra)
    assert cli.list_processes(ap.parse_args(['--image', 'a.fits']), 1) == 1
    monkeypatch.setenv('BBX_CPU_BUDGET', '16')
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '1')
    files = ['/data/ML1_%02d.fits.fz' % k for k in range(7)]
    started = []

    class FakeChild:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            self.cmd, self.env = cmd, env
            lst = cmd[cmd.index('--image_list') + 1]
            self.share = [ln.strip() for ln in open(lst) if ln.strip()]
            self.k = len(started)
            started.append(self)
            self.returncode = None

        def communicate(self):
            if self.k == 1:                                       # the second child dies half way
                self.returncode = 1
                return 'some line\n', None
            self.returncode = 0
            out = ['INFO chatter'] + [f.replace('.fits.fz', '_red.fits.fz') if not f.endswith('03.fits.fz') else 'None' for f in self.share]
            out.append('BBX_TIMING ' + json.dumps(dict(marks=[['calibration_and_reference_files_in_hbm', 1.5]], t_module_import_unix=100.0,
                                                       files_done_unix=[101.0 + i for i in range(len(self.share))], pipeline=dict(frames=len(self.share)),
                                                       hbm_peak_GB_tensors=50.0)))
            return '\n'.join(out) + '\n', None

        def poll(self):
            return self.returncode

        def kill(self):
            pass
    monkeypatch.setattr(subprocess, 'Popen', FakeChild)
    monkeypatch.setenv('BBX_TIMING', '1')
    argv = ['--telescope', 'ML1', '--image_list', 'whatever.txt', '--fpack', 'True', '--list_procs=3', '--red_dir', str(tmp_path)]
    res = cli.reduce_list_in_processes(argv, files, 3)
    assert [c.share for c in started] == [files[0::3], files[1::3], files[2::3]]
    for c in started:
        tail = c.cmd[2:]
        assert tail[:6] == ['--telescope', 'ML1', '--fpack', 'True', '--red_dir', str(tmp_path)] and tail[-2:] == ['--list_procs', '1']
        assert tail.count('--image_list') == 1 and c.env['BBX_CPU_BUDGET'] == '5' and c.env['BBX_TIMING'] == '1'
    want = [f.replace('.fits.fz', '_red.fits.fz') for f in files]
    want[3] = None                                                # reported None by its (healthy) child
    for k in (1, 4):
        want[k] = None                                            # the files of the child that died
    assert res == want
    printed = capsys.readouterr().out.splitlines()
    assert printed[:7] == [str(w) for w in want]
    tm = json.loads([ln for ln in printed if ln.startswith('BBX_TIMING ')][-1][len('BBX_TIMING '):])
    assert tm['list_processes'] == 3 and len(tm['files_done_unix']) == 5 and tm['hbm_peak_GB_tensors'] == 100.0
