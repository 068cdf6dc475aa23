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
