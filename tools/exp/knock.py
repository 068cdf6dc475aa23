"""GPU box: headline frames/s with whole entry points knocked out (NAME: returns BBX_OK without launching anything) or
doubled (dup:NAME: called twice, for entry points whose outputs do not depend on being called once): what a stage is worth
to the pipeline's rate.  Results may be garbage; only the clock is read.
    python3 tools/exp/knock.py [name,name,...] [bench args]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))


def twice(f):
    def g(*a):
        rc = f(*a)
        return rc if rc else f(*a)
    return g


if __name__ == '__main__':                                       # (the fit workers are spawned: they import this file)
    names = [n for n in sys.argv[1].split(',') if n and n != 'none']
    from blackbox_amd import _lib
    for n in names:
        dup = n.startswith('dup:')
        n = n[4:] if dup else n
        assert hasattr(_lib.lib, n), n
        setattr(_lib.lib, n, twice(getattr(_lib.lib, n)) if dup else (lambda *a, **k: 0))
    import bench
    sys.argv = ['bench.py', '--no-cpu', '--no-extras'] + sys.argv[2:]
    bench.main()
