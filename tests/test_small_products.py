"""CPU: the small products of a frame (mini images, catalogue tables, header files) as one task for a worker process
(blackbox_amd/catalogs.write_small_products; the list run of blackbox.py hands them to the host pool so that their formatting
does not hold the interpreter lock of the process that drives the GPU): the files are the ones the direct calls make, and the
task survives the trip through a process pool (pickling of headers, tables and numpy arrays)."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from blackbox_amd import catalogs, fitsio


def _jobs(base):
    rs = np.random.RandomState(3)
    hdr = {'EXPTIME': (60.0, '[s] exposure time'), 'OBJECT': 'field 1', 'QC-FLAG': ('green', 'QC flag'), 'S-BKGSTD': (np.float32(9.5), 'sigma'),
           'NOBJECTS': (np.int64(12), 'objects'), 'Z-P': (True, 'ZOGY?')}
    cat = dict(Y_POS=rs.rand(12).astype(np.float32) * 100, X_POS=rs.rand(12).astype(np.float32) * 100, E_FLUX_PEAK=rs.rand(12).astype(np.float32),
               E_FLUX_OPT=rs.rand(12).astype(np.float32), E_FLUXERR_OPT=rs.rand(12).astype(np.float32), SNR_OPT=rs.rand(12).astype(np.float32))
    trans = [dict(y=3, x=4, scorr=7.5, fpsf=100.0, fpsferr=12.0), dict(y=30, x=41, scorr=-8.25, fpsf=-90.0, fpsferr=11.0)]
    mini = rs.rand(6, 7).astype(np.float32)
    return [('image', (base + '_bkg_mini.fits', mini, {'BKG-SIZE': (60, 'box')})), ('cat', (cat, base + '_cat.fits', 'new', dict(hdr))),
            ('header', (base + '_cat_hdr.fits', dict(hdr))), ('trans', (trans, base + '_trans.fits', dict(hdr, **{'T-NTRANS': (2, 'n')}))),
            ('header', (base + '_hdr.fits', dict(hdr)))], (hdr, cat, trans, mini)


def test_one_task_writes_what_the_direct_calls_write(tmp_path):
    a, b = str(tmp_path / 'a'), str(tmp_path / 'b')
    jobs, (hdr, cat, trans, mini) = _jobs(a)
    done = catalogs.write_small_products(jobs)
    assert done == [a + s for s in ('_bkg_mini.fits', '_cat.fits', '_cat_hdr.fits', '_trans.fits', '_hdr.fits')]
    fitsio.write_image(b + '_bkg_mini.fits', mini, {'BKG-SIZE': (60, 'box')})
    catalogs.format_cat(cat, b + '_cat.fits', cat_type='new', header2add=hdr)
    fitsio.write_header(b + '_cat_hdr.fits', hdr)
    catalogs.format_cat(catalogs.transient_table(trans), b + '_trans.fits', cat_type='trans', header2add=dict(hdr, **{'T-NTRANS': (2, 'n')}))
    fitsio.write_header(b + '_hdr.fits', hdr)
    for s in ('_bkg_mini.fits', '_cat.fits', '_cat_hdr.fits', '_trans.fits', '_hdr.fits'):
        assert open(a + s, 'rb').read() == open(b + s, 'rb').read(), s
    assert os.path.getsize(a + '_cat.fits') % 2880 == 0
    with pytest.raises(ValueError):
        catalogs.write_small_products([('nonsense', ('x',))])


def test_the_task_crosses_a_process_pool(tmp_path):
    a, b = str(tmp_path / 'a'), str(tmp_path / 'b')
    jobs_a, _ = _jobs(a)
    jobs_b, _ = _jobs(b)
    catalogs.write_small_products(jobs_a)
    got, err = [], []
    with mp.get_context('spawn').Pool(1) as pool:
        r = pool.apply_async(catalogs.write_small_products, (jobs_b,), callback=got.append, error_callback=err.append)
        r.wait(120)
        # a task that fails reports through the error callback (the frame's file group then carries the error)
        r2 = pool.apply_async(catalogs.write_small_products, ([('image', (str(tmp_path / 'no_such_dir' / 'x.fits'), np.zeros((2, 2), np.float32), {}))],),
                              callback=got.append, error_callback=err.append)
        r2.wait(120)
    assert len(got) == 1 and len(got[0]) == 5 and len(err) == 1 and isinstance(err[0], OSError)
    for s in ('_bkg_mini.fits', '_cat.fits', '_cat_hdr.fits', '_trans.fits', '_hdr.fits'):
        assert open(a + s, 'rb').read() == open(b + s, 'rb').read(), s
