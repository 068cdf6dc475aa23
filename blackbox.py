#!/usr/bin/env python
"""blackbox.py -- per-file entry point of the MI355X reduction, same command line as the
reference's `python blackbox.py --telescope T --img_reduce True --cat_extract True
--trans_extract True --force_reproc_new True --image FILE` (blackbox.py:8128-8213,
blackbox_slurm_google.py:305-309).

FITS in -> `{tel}_{yyyymmdd}_{hhmmss}_red.fits` (float32, e-), `..._mask.fits` (uint8),
`..._red_hdr.fits`, `..._red.log`; with --cat_extract / --trans_extract the products of
zogy.optimal_subtraction that the hot path covers (set_blackbox.py:157-164): `_red_bkg_mini`,
`_red_bkg_std_mini`, `_red_cat` (+ `_red_cat_hdr`), `_red_D`, `_red_Scorr`, `_red_Fpsf`,
`_red_trans` (+ `_red_trans_hdr`).  The reference image, its mask and the PSFs are explicit
inputs (--ref, --ref_mask, --psf_new, --psf_ref), like the masters (--mflat, --mbias, --bpm):
their selection / creation (reference building, PSFEx, astrometry) is orchestration.
Flags of the reference that concern orchestration only (night mode, dates, master creation)
are accepted and ignored with a warning.  With WORLD_SIZE > 1 (torch.distributed.run) every
rank takes every world-th file of --image_list on its own GPU; an --image_list of object
frames runs through the frames-in-flight pipeline (blackbox_amd.pipeline.FramePipeline).
"""
import argparse
import logging
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KEYWORDS_VERSION = '1.2.2'      # header keywords version this product follows (blackbox.py:123)
log = logging.getLogger('blackbox')

# BBX_TIMING=1: where the wall time of a `python blackbox.py --image F` process goes (the reference's Slurm contract is one
# interpreter per file, blackbox_slurm_google.py:305-309): seconds since this module was imported at each mark, printed
# as one `BBX_TIMING {json}` line when main() returns
_T0 = time.time()
_MARKS = []
_PIPE_STATS = {}          # --image_list: mean wall time of a frame per pipeline state (BBX_TIMING)
_FILES_DONE = []          # --image_list: the moment every product of a file was on disk (BBX_TIMING: the list's rate)


def _mark(name):
    _MARKS.append((name, time.time() - _T0))


def _start_sampler(path):
    """BBX_CLI_SAMPLE=<file> (debug): every 5 ms the innermost Python frame of every thread; at exit the most frequent
    (thread, function) pairs go to the file -- where the threads of a list run spend their wall time (a thread inside a
    library call that released the interpreter lock shows the calling line; one waiting for the lock shows wherever it
    stopped)"""
    if not path:
        return None
    import atexit
    import collections
    import threading
    counts = collections.Counter()
    stop = threading.Event()

    def run():
        me = threading.get_ident()
        while not stop.wait(0.005):
            names = {t.ident: t.name for t in threading.enumerate()}
            for tid, fr in sys._current_frames().items():
                if tid == me:
                    continue
                name = names.get(tid, '?').rstrip('0123456789-_ ')
                chain = []
                f = fr
                while f is not None and len(chain) < 3:
                    chain.append('%s:%s:%d' % (os.path.basename(f.f_code.co_filename), f.f_code.co_name, f.f_lineno))
                    f = f.f_back
                counts[(name, ' < '.join(chain))] += 1
    th = threading.Thread(target=run, daemon=True, name='sampler')
    th.start()

    def dump():
        stop.set()
        per = collections.Counter()
        for (name, _), c in counts.items():
            per[name] += c
        with open(path, 'w') as f:
            for name, tot in per.most_common():
                f.write('== %s: %d samples\n' % (name, tot))
                for (n2, where), c in counts.most_common():
                    if n2 == name and c >= 0.02 * tot:
                        f.write('   %5.1f %%  %s\n' % (100.0 * c / tot, where))
    atexit.register(dump)
    return th


def str2bool(v):
    """blackbox.py:8115-8123"""
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


class WrapException(Exception):
    """the exception of a worker with its formatted traceback, so that a multiprocessing pool can hand it to
    the parent (what blackbox.py:933-943 is for): raised by try_blackbox_reduce in place of whatever
    blackbox_reduce raised"""

    def __init__(self):
        import traceback
        exc_type, exc_value, exc_tb = sys.exc_info()
        Exception.__init__(self, repr(exc_value))
        self.exception = exc_value
        self.formatted = ''.join(traceback.format_exception(exc_type, exc_value, exc_tb))

    def __reduce__(self):                                          # crosses the pool as text
        return (_rebuild_wrapexception, (self.args, self.formatted))

    def __str__(self):
        return '{}\nOriginal traceback:\n{}'.format(Exception.__str__(self), self.formatted)


def _rebuild_wrapexception(args, formatted):
    e = WrapException.__new__(WrapException)
    Exception.__init__(e, *args)
    e.exception, e.formatted = None, formatted
    return e


# ---- operator-level entry the reference's farm imports (blackbox.py:363-379: `pool_func(try_blackbox_reduce,
# filenames, nproc)`): module-level functions of one argument.  The settings the reference keeps in module globals
# set by run_blackbox (tel, filts, types, proc_mode; blackbox.py:141, 321-323) are the parsed command line here:
# configure() stores it in this process and in the environment, so that spawned pool workers -- which import this
# module afresh and must create their GPU context themselves -- find it.
_ENV_KEY = 'BBX_BLACKBOX_ARGV'
_STATE = {'argv': None, 'reducer': None}


def configure(argv):
    """set the per-process settings (the command-line flags of main(), as a list of strings) for blackbox_reduce /
    try_blackbox_reduce; inherited by pool workers through the environment.  Makes no GPU call."""
    import json
    _STATE['argv'] = list(argv)
    _STATE['reducer'] = None
    os.environ[_ENV_KEY] = json.dumps(_STATE['argv'])


def _reducer(filename=None):
    """this process's Reducer (GPU context + masters resident in HBM), created at the first frame"""
    if _STATE['reducer'] is None:
        if _STATE['argv'] is None:
            import json
            if _ENV_KEY not in os.environ:
                raise RuntimeError('blackbox.configure(argv) has not been called (and %s is not set)' % _ENV_KEY)
            _STATE['argv'] = json.loads(os.environ[_ENV_KEY])
        args = build_parser().parse_args(_STATE['argv'])
        _STATE['reducer'] = Reducer(telescope_of(args, filename), args)
    return _STATE['reducer']


def blackbox_reduce(filename):
    """blackbox.py:1027-2669 for one file: -> path of the reduced image, or None if it was skipped"""
    return _reducer(filename).blackbox_reduce(filename)


def try_blackbox_reduce(filename):
    """blackbox.py:948-999: blackbox_reduce in a try / except that re-raises as WrapException (formatted traceback
    for the parent of a multiprocessing pool); the per-image log is closed in blackbox_reduce's own finally"""
    try:
        return blackbox_reduce(filename)
    except BaseException:
        raise WrapException()


def pool_func(func, filelist, nproc=1):
    """blackbox.py:363-379 / zogy.pool_func: map [func] over the files with [nproc] worker processes.  Workers
    are SPAWNED (a forked child cannot use a GPU runtime its parent initialised) and each creates its own GPU
    context at its first file; the parent makes no GPU call.  WORLD_SIZE / LOCAL_RANK are not needed: worker k
    takes GPU k modulo the number of visible devices."""
    import multiprocessing as mp
    if nproc <= 1:
        return [func(f) for f in filelist]
    with mp.get_context('spawn').Pool(nproc, initializer=_pool_worker_init, initargs=(os.environ.get(_ENV_KEY),)) as pool:
        return pool.map(func, filelist, chunksize=1)


def _pool_worker_init(argv_json):
    import multiprocessing as mp
    if argv_json is not None:
        os.environ[_ENV_KEY] = argv_json
    ident = mp.current_process()._identity
    import torch
    ndev = max(1, torch.cuda.device_count())                      # counting devices does not initialise the GPU
    os.environ['LOCAL_RANK'] = str(((ident[0] - 1) if ident else 0) % ndev)
    os.environ['RANK'], os.environ['WORLD_SIZE'] = '0', '1'      # the pool hands out the files; no sharding inside a worker


def telescope_of(args, filename=None):
    """telescope from the file name prefix, else --telescope (blackbox.py:145-152)"""
    tel = args.telescope
    if filename:
        base = os.path.basename(filename)
        for t in ('ML1', 'BG2', 'BG3', 'BG4'):
            if base.startswith(t):
                tel = t
    return tel


def outname(header, tel, red_dir):
    """{tel}_{yyyymmdd}_{hhmmss}_red.fits from DATE-OBS (blackbox.py:1004-1022, 1166-1184)"""
    from blackbox_amd.reduce import hval
    date_obs = str(hval(header, 'DATE-OBS')) if 'DATE-OBS' in header else time.strftime('%Y-%m-%dT%H:%M:%S')
    d, t = date_obs.split('T')[0].replace('-', ''), date_obs.split('T')[-1].split('.')[0].replace(':', '')
    return os.path.join(red_dir, '{}_{}_{}_red.fits'.format(tel, d, t))


def _base(p):
    return os.path.basename(p) if p else 'None'


def _stem(p):
    return os.path.basename(p).split('.fits')[0] if p else 'None'


class Reducer:
    """per-process state: GPU context + masters / reference / PSFs resident in HBM"""

    def __init__(self, tel, args):
        _mark('main_start')
        import torch
        _mark('torch_imported')
        from blackbox_amd import fitsio, reduce as R
        _mark('package_imported')
        self.R, self.torch, self.fitsio = R, torch, fitsio
        _, _, local = __import__('blackbox_amd.farm', fromlist=['rank_world']).rank_world()
        if os.environ.get('BBX_ONE_GPU'):                          # rehearsal of a multi-rank run on a one-GPU box: every rank on device 0
            local = 0
        self.ctx = R.Context(local)
        _mark('gpu_context')
        self.tel = tel
        self.args = args
        dev = self.ctx.device

        def load(path, dtype, what):
            """a master that cannot be read is not applied: <X>-P False (blackbox.py:1674-1699, 1820-1846)"""
            if not path:
                return None
            try:
                return R.image_to_device(self.ctx, path, dtype)      # (float32 files: bytes up, swapped on the device)
            except Exception:
                log.exception('exception was raised while reading the %s %s', what, path)
                return None
        self.mflat = load(args.mflat, np.float32, 'master flat')
        self.mbias = load(args.mbias, np.float32, 'master bias')
        self.bpm = load(args.bpm, np.uint8, 'bad pixel mask')
        if self.bpm is not None and bool((self.bpm & 12).any()):
            raise ValueError('bad-pixel mask carries saturated(4)/saturated-connected(8) bits; expected 1 and 32 only')
        self.xtalk = None
        if args.crosstalk:
            try:
                self.xtalk = R.read_crosstalk(args.crosstalk)
            except Exception:
                log.exception('exception was raised while reading the crosstalk file %s', args.crosstalk)
        self.nonlin = None
        if args.nonlin:
            try:
                self.nonlin = R.read_nonlin_splines(args.nonlin)
            except Exception:
                log.exception('exception was raised while reading the non-linearity splines %s', args.nonlin)
        # inputs of the subtraction stage
        self.sub = None
        if args.cat_extract or args.trans_extract:
            self.sub = self._load_subtraction_inputs(load)
        _mark('calibration_and_reference_files_in_hbm')

    def _load_psf(self, path):
        """PSF input: a PSFEx .psf table (model evaluated on the GPU) or a FITS image / cube of
        unit-sum stamps ([S, S] or [nsub, S, S])"""
        if not path:
            return None
        torch, fitsio = self.torch, self.fitsio
        try:
            if path.endswith('.psf') or path.endswith('_psf.fits'):
                m = fitsio.read_psfex(path)
                if m['psf_samp'] != 1.0:
                    # the model is tabulated every PSF_SAMP pixels: bring the basis to image pixels (zogy.get_psf_ima)
                    from blackbox_amd import zogy as G
                    if not (0.05 <= m['psf_samp'] <= 20.0):
                        raise ValueError('PSF_SAMP {} out of range'.format(m['psf_samp']))
                    m['basis'] = G.resample_psf_basis(m['basis'], m['psf_samp'])
                    log.info('PSF model resampled from PSF_SAMP %.3f to image pixels: %d x %d', m['psf_samp'], *m['basis'].shape[1:])
                m['basis'] = torch.from_numpy(np.ascontiguousarray(m['basis'])).to(self.ctx.device)
                return m
            st = np.ascontiguousarray(fitsio.read_image(path, dtype=np.float32))
            st = st / st.sum(axis=(-2, -1), keepdims=True)
            return torch.from_numpy(st).to(self.ctx.device)
        except Exception:
            log.exception('exception was raised while reading the PSF %s', path)
            return None

    def _load_subtraction_inputs(self, load):
        a = self.args
        sub = dict(psf_new=self._load_psf(a.psf_new), psf_ref=None, ref=None, ref_mask=None,
                   cat_extract=bool(a.cat_extract), trans_extract=bool(a.trans_extract),
                   fratio=a.fratio, dx=a.zogy_dx, dy=a.zogy_dy, ref_is_bkgsub=bool(a.ref_bkgsub))
        if a.subimage_size:
            sub['subimage_size'] = a.subimage_size
        if a.subimage_border is not None:
            sub['subimage_border'] = a.subimage_border
        if a.bkg_boxsize:
            sub['bkg_boxsize'] = a.bkg_boxsize
        if a.trans_extract and a.ref:
            sub['ref'] = load(a.ref, np.float32, 'reference image')
            sub['ref_mask'] = load(a.ref_mask, np.uint8, 'reference mask') if a.ref_mask else None
            if sub['ref'] is not None and sub['ref_mask'] is None:
                sub['ref_mask'] = self.torch.zeros(sub['ref'].shape, dtype=self.torch.uint8, device=self.ctx.device)
            sub['psf_ref'] = self._load_psf(a.psf_ref)
            if a.ref_bkg_std_mini:
                sub['ref_bkg_std_mini'] = self.fitsio.read_image(a.ref_bkg_std_mini, dtype=np.float32)
            if sub['ref'] is None or sub['psf_ref'] is None or sub['psf_new'] is None:
                log.error('reference image or PSFs missing: processing the new image only')
                sub['ref'] = None
        elif a.trans_extract:
            log.info('no reference image given: processing the new image only, without comparison to a reference '
                     '(blackbox.py:2338-2354)')
        return sub

    # ------------------------------------------------------------------------------------
    def try_blackbox_reduce(self, filename):
        """blackbox.py:948-999: the reduced file name or None (skipped); raises WrapException with the formatted
        traceback when blackbox_reduce raised"""
        try:
            return self.blackbox_reduce(filename)
        except BaseException:
            raise WrapException()

    def reduce_logged(self, filename):
        """what the reference's single-process loop amounts to for the caller of main(): a failing file is logged
        with its traceback and reported as None, the run goes on"""
        try:
            return self.try_blackbox_reduce(filename)
        except WrapException as e:
            log.error('exception was raised during [blackbox_reduce] of %s:\n%s', filename, e.formatted)
            return None

    def read_raw(self, filename):
        """-> (raw device tensor, header dict)"""
        R, torch, fitsio = self.R, self.torch, self.fitsio
        if filename.endswith('.fz'):
            # fpacked raw frame (the reference's usual input): the compressed bytes go to the GPU
            # and are decoded there
            from blackbox_amd import fpack as P
            d_raw, hraw = P.funpack_image(self.ctx, filename)
            if d_raw.dtype not in (torch.uint16, torch.float32):
                d_raw = d_raw.to(torch.float32)
        else:
            raw, hraw = fitsio.read_image(filename, get_header=True)
            if raw.dtype not in (np.uint16, np.float32):
                raw = raw.astype(np.float32)
            d_raw = torch.from_numpy(np.ascontiguousarray(raw)).to(self.ctx.device)
        header = dict(hraw)
        for k in ('BZERO', 'BSCALE', 'BITPIX', 'NAXIS', 'NAXIS1', 'NAXIS2', 'SIMPLE', 'EXTEND'):
            header.pop(k, None)
        return d_raw, header

    def names(self, header):
        red_dir = self.args.red_dir or '.'
        fits_out = outname(header, self.tel, red_dir)
        return red_dir, fits_out

    def already_done(self, fits_out):
        return os.path.isfile(fits_out) and os.path.isfile(fits_out.replace('_red', '_mask')) \
            and not (self.args.img_reduce and self.args.force_reproc_new)

    def blackbox_reduce(self, filename):
        R = self.R
        t0 = time.time()
        d_raw, header = self.read_raw(filename)
        _mark('raw_read')
        if not self.args.red_dir:
            self.args.red_dir = os.path.dirname(os.path.abspath(filename))
        red_dir, fits_out = self.names(header)
        if self.already_done(fits_out):
            log.info('%s already reduced; skipping', filename)           # blackbox.py:1336-1390
            return fits_out
        os.makedirs(red_dir, exist_ok=True)
        fh = self.open_image_log(fits_out)
        try:
            exptime = R.hval(header, 'EXPTIME') if 'EXPTIME' in header else 1.0
            imgtype = str(R.hval(header, 'IMAGETYP')).lower() if 'IMAGETYP' in header else 'object'
            data, mask, header, hm = R.reduce_object(
                self.ctx, d_raw, header, self.tel, mflat=self.mflat, mbias=self.mbias, bpm=self.bpm,
                xtalk_coeffs=self.xtalk, exptime=exptime, ysize_chan=self.args.ysize_chan,
                xsize_chan=self.args.xsize_chan, nonlin_splines=self.nonlin, imgtype=imgtype, log=log)
            _mark('reduced')
            if imgtype != 'object':
                return self.finish_calibration_frame(data, mask, header, imgtype, fits_out)
            return self.finish_object(filename, data, mask, header, hm, fits_out, t0)
        finally:
            self.close_image_log(fh)

    # ---- per-image log (blackbox.py:1311-1318: {base}_red.log next to the product) ----------
    def open_image_log(self, fits_out):
        try:
            fh = logging.FileHandler(fits_out.replace('.fits', '.log'), mode='w')
            fh.setFormatter(logging.Formatter('%(asctime)s [%(levelname)s, %(process)s] %(message)s'))
            log.addHandler(fh)
            return fh
        except OSError:
            return None

    def close_image_log(self, fh):
        if fh is not None:
            log.removeHandler(fh)
            fh.close()

    # ---- bias / dark / flat frames -------------------------------------------------------------
    def finish_calibration_frame(self, data, mask, header, imgtype, fits_out):
        """blackbox.py:1627-1637 (bias), 1764-1784 (flat): statistics, QC flags, write the frame
        (no mask product)"""
        R, fitsio = self.R, self.fitsio
        from blackbox_amd import qc
        header['BUNIT'] = ('e-', 'pixel values are in electrons')
        if imgtype == 'flat' and R.hval(header, 'OS-P'):
            from blackbox_amd import flatstats
            flatstats.get_flatstats(self.ctx, data, header, mask, self.tel, ysize_chan=self.args.ysize_chan,
                                    xsize_chan=self.args.xsize_chan)
        self.bookkeeping(header, time.time())
        qc.run_qc_check(header, self.tel)
        fits_out = fits_out.replace('_red.fits', '.fits')       # calibration frames keep their plain name
        fitsio.write_image(fits_out, data.cpu().numpy(), header)
        return fits_out

    def bookkeeping(self, header, t0):
        """bookkeeping keywords of the header contract (blackbox.py:1110-1112, 1520, 1607, 1669-1690,
        1817-1855; verify_header 2893-3255)"""
        R = self.R
        from blackbox_amd import __version__ as bbx_version
        a = self.args
        header['BB-V'] = ('amd-' + bbx_version, 'BlackBOX version used')
        header['KW-V'] = (KEYWORDS_VERSION, 'header keywords version used')
        header['BB-START'] = (time.strftime('%Y-%m-%dT%H:%M:%S', time.gmtime(t0)), 'start UTC date of BlackBOX image run')
        header['XTALK-F'] = (_base(a.crosstalk), 'name crosstalk coefficients file')
        header.setdefault('GAIN', (1.0, '[e-/ADU] effective gain all channels'))
        header.setdefault('GAIN-P', (True, 'corrected for gain?'))
        header.setdefault('NONLIN-P', (False, 'corrected for non-linearity?'))
        header['NONLIN-F'] = (_base(a.nonlin) if R.hval(header, 'NONLIN-P') else 'None', 'name non-linearity correction file')
        header['MBIAS-F'] = (_stem(a.mbias) if header.get('MBIAS-P') and R.hval(header, 'MBIAS-P') else 'None',
                             'name of master bias applied')
        header['MFLAT-F'] = (_stem(a.mflat) if header.get('MFLAT-P') and R.hval(header, 'MFLAT-P') else 'None',
                             'name of master flat applied')
        for k, c in (('XTALK-P', 'corrected for crosstalk?'), ('COSMIC-P', 'corrected for cosmic rays?'),
                     ('SAT-P', 'processed for satellite trails?'), ('MBIAS-P', 'corrected for master bias?'),
                     ('MFLAT-P', 'corrected for master flat?'), ('MASK-P', 'mask image created?')):
            header.setdefault(k, (False, c))                                     # steps that did not run
        header.setdefault('NCOSMICS', ('None', '[/s] number of cosmic rays identified'))
        header.setdefault('NSATS', ('None', 'number of satellite trails identified'))
        header['MFRING-P'] = (False, 'corrected for master fringe map?')
        header['MFRING-F'] = ('None', 'name of master fringe map applied')
        header['FRRATIO'] = ('None', 'fringe ratio (science/fringe map) applied')

    # ---- object frames -------------------------------------------------------------------------
    def finish_object(self, filename, data, mask, header, hm, fits_out, t0, sub_result=None):
        R, fitsio = self.R, self.fitsio
        from blackbox_amd import qc
        header['BUNIT'] = ('e-', 'pixel values are in electrons')
        self.bookkeeping(header, t0)
        redfile = os.path.basename(fits_out).split('.fits')[0]
        header['REDFILE'] = (redfile, 'BlackBOX reduced image name')
        header['MASKFILE'] = (redfile.replace('_red', '_mask'), 'BlackBOX mask image name')
        # QC flags on the reduction keywords (blackbox.py:2000; qc.py)
        qc_flag = qc.run_qc_check(header, self.tel, check_key_type='full')      # flags go into the image header
        qc.verify_header(header, ['full'], name=fits_out)                     # blackbox.py:2062 (raises on a DB keyword)
        base = fits_out.replace('.fits', '')
        if qc_flag == 'red':
            # red flag: dummy catalogues, no subtraction (blackbox.py:2015-2046)
            log.error('red QC flag in image %s; making dummy catalogs and returning', fits_out)
            written = self.write_image(fits_out, data, header)
            self.write_image(fits_out.replace('_red', '_mask'), mask, hm)
            fitsio.write_header(base + '_hdr.fits', header)
            if self.sub is not None:
                qc.run_qc_check(header, self.tel, cat_type='new', cat_dummy=base + '_cat.fits', check_key_type='full')
                htrans = dict(header)
                qc.run_qc_check(htrans, self.tel, cat_type='trans', cat_dummy=base + '_trans.fits', check_key_type='trans')
            return written
        if self.sub is not None:
            self.subtract_and_write(data, mask, header, base, sub_result)
        written = self.write_image(fits_out, data, header)
        self.write_image(fits_out.replace('_red', '_mask'), mask, hm)
        self._small('header', base + '_hdr.fits', dict(header))               # update_imcathead(create_hdrfile=True), 2011
        log.info('reduced %s -> %s in %.2f s', filename, written, time.time() - t0)
        _mark('products_written')
        return written

    def _small(self, kind, *args):
        """a small product (mini image, table, header file): written at once -- or, for a frame of the list run whose
        images the output stage is writing, noted for one call of catalogs.write_small_products in a worker process"""
        staged = getattr(self, '_staged', None)
        if staged is not None and staged.get('deferred') is not None:
            staged['deferred'].append((kind, args))
            return
        from blackbox_amd import catalogs
        catalogs.write_small_products([(kind, args)])

    def write_image(self, path, img, header):
        staged = getattr(self, '_staged', None)
        if staged is not None and (path + '.fz') in staged['names']:
            # the output stage is writing this image already (compressed on its lane): it only waits for the header
            staged['headers'][path + '.fz'] = dict(header)
            staged['wanted'].add(path + '.fz')
            return path + '.fz'
        if self.args.fpack:
            # products leave the GPU tile-compressed (reference: fpack of the files kept,
            # copy_files2keep 4033-4035): only the compressed bytes cross PCIe
            from blackbox_amd import fpack as P
            if self.torch.is_tensor(img):
                P.fpack_image(self.ctx, path, img, header)
                return path + '.fz'
        self.fitsio.write_image(path, img.cpu().numpy() if self.torch.is_tensor(img) else img, header)
        return path

    def subtract_and_write(self, data, mask, header, base, res=None):
        """zogy.optimal_subtraction + the products it leaves (set_blackbox.py:157-164); a failure
        keeps the reduction products (blackbox.py:2364-2382)"""
        from blackbox_amd import zogy as G, qc
        fitsio = self.fitsio
        try:
            if res is None:
                kw = {k: v for k, v in self.sub.items() if k not in ('ref', 'ref_mask', 'psf_new', 'psf_ref')}
                res = G.optimal_subtraction(self.ctx, data, self.sub['ref'], mask, self.sub['ref_mask'],
                                            self.sub['psf_new'], self.sub['psf_ref'], **kw)
                self.ctx.sync()
                _mark('subtracted')
        except Exception:
            log.exception('exception was raised during [optimal_subtraction]; saving just the image reduction products')
            header['Z-P'] = (False, 'successfully processed by ZOGY?')
            return
        hnew, htrans = res['header_new'], res['header_trans']
        header.update(hnew)
        bkg_hdr = {'BKG-SIZE': hnew['BKG-SIZE']}
        self._small('image', base + '_bkg_mini.fits', np.asarray(res['bkg_mini_new']), bkg_hdr)
        self._small('image', base + '_bkg_std_mini.fits', np.asarray(res['bkg_std_mini_new']), bkg_hdr)
        qc_flag = qc.run_qc_check(header, self.tel, check_key_type='full')
        if res.get('catalog') is not None:
            if qc_flag == 'red':
                qc.run_qc_check(header, self.tel, cat_type='new', cat_dummy=base + '_cat.fits', check_key_type='full')
            else:
                self._small('cat', res['catalog'], base + '_cat.fits', 'new', dict(header))
            self._small('header', base + '_cat_hdr.fits', dict(header))
        if res.get('D') is not None:
            full_t = dict(header)
            full_t.update(htrans)
            tqc = qc.run_qc_check(full_t, self.tel, check_key_type='trans')
            for ext in ('D', 'Scorr', 'Fpsf'):
                self.write_image('{}_{}.fits'.format(base, ext), res[ext], full_t)
            self.write_limmag(base, res, full_t)
            if tqc == 'red' or qc_flag == 'red':
                qc.run_qc_check(full_t, self.tel, cat_type='trans', cat_dummy=base + '_trans.fits', check_key_type='trans')
            else:
                self._small('trans', res['transients'], base + '_trans.fits', dict(full_t))
            self._small('header', base + '_trans_hdr.fits', dict(full_t))

    def _limmag_is_flux(self, header):
        """one rule for the unit of `_trans_limmag`, whichever path writes it: a flux limit only when no zeropoint is known --
        neither --zeropoint nor a numeric PC-ZP in the frame's header (then the output stage may queue it on the lane)"""
        if self.args.zeropoint is not None:
            return False
        return not ('PC-ZP' in header and not isinstance(self.R.hval(header, 'PC-ZP'), str))

    def write_limmag(self, base, res, header):
        """`_trans_limmag.fits` (set_blackbox.py:160-162): the transient detection limit per pixel,
        T-NSIGMA x Fpsferr -- in magnitudes when a zeropoint is known (--zeropoint, or PC-ZP of the header, with the
        extinction term PC-EXTCO x AIRMASSC when both are there: the photometric calibration itself is outside this
        path), else as a flux in e- (LIMUNIT says which)"""
        R, torch = self.R, self.torch
        if res.get('Fpsferr') is None:
            return
        nsig = float(R.hval(header, 'T-NSIGMA')) if 'T-NSIGMA' in header else 6.0
        zp = self.args.zeropoint
        staged = getattr(self, '_staged', None)
        is_staged = staged is not None and (base + '_trans_limmag.fits.fz') in staged['names']
        # (the output stage has queued the flux limit already: only its header is made here, no image)
        lim = None if is_staged else res['Fpsferr'] * nsig
        if is_staged:
            zp = None
        elif zp is None and 'PC-ZP' in header and not isinstance(R.hval(header, 'PC-ZP'), str):
            zp = float(R.hval(header, 'PC-ZP'))
        h = dict(header)
        if zp is not None:
            exptime = float(R.hval(header, 'EXPTIME')) if 'EXPTIME' in header else 1.0
            ext = 0.0
            if 'PC-EXTCO' in header and 'AIRMASSC' in header:
                try:
                    ext = float(R.hval(header, 'PC-EXTCO')) * float(R.hval(header, 'AIRMASSC'))
                except (TypeError, ValueError):
                    ext = 0.0
            lim = torch.where(lim > 0, zp - 2.5 * torch.log10(lim.clamp(min=1e-30) / exptime) - ext, torch.zeros_like(lim))
            h['LIMUNIT'] = ('mag', 'unit of the limiting-magnitude image')
        else:
            h['LIMUNIT'] = ('e-', 'limit in flux: no zeropoint was given')
        h['LIMNSIG'] = (nsig, '[sigma] significance of the limit')
        self.write_image(base + '_trans_limmag.fits', lim.contiguous() if lim is not None else None, h)

    # ---- many object frames: frames-in-flight pipeline ------------------------------------------
    def reduce_list(self, files):
        """object frames of one geometry through FramePipeline (several frames in flight on this
        GPU); anything else goes through blackbox_reduce one by one.  Raw frames are read and uploaded as
        the pipeline takes them (at most its depth in HBM at a time).  With --fpack True the image products are
        compressed on the lane that made them and written by the output stage's threads
        (blackbox_amd/outstage.py).  -> list of output names"""
        R, torch = self.R, self.torch
        from blackbox_amd.pipeline import FramePipeline
        out = {}
        todo = []
        for fn in files:
            try:
                header = self.read_header(fn)
                imgtype = str(R.hval(header, 'IMAGETYP')).lower() if 'IMAGETYP' in header else 'object'
                if not self.args.red_dir:
                    self.args.red_dir = os.path.dirname(os.path.abspath(fn))
                _, fits_out = self.names(header)
                if imgtype != 'object' or self.nonlin is not None:
                    out[fn] = self.reduce_logged(fn)
                elif self.already_done(fits_out):
                    out[fn] = fits_out
                else:
                    todo.append((fn, fits_out))
            except Exception:
                log.exception('exception was raised while reading %s', fn)
                out[fn] = None
        if not todo:
            return [out.get(fn) for fn in files]
        # the list's geometry from its first readable file (one that is not fails alone, like any other: blackbox.py:948-999)
        first_raw = None
        while todo and first_raw is None:
            try:
                first_raw, first_hdr = self.read_raw(todo[0][0])
            except Exception:
                log.exception('exception was raised while reading %s', todo[0][0])
                out[todo.pop(0)[0]] = None
        if not todo:
            return [out.get(fn) for fn in files]
        geom = R.geometry(first_raw.shape, self.args.ysize_chan, self.args.xsize_chan)
        exptime = R.hval(first_hdr, 'EXPTIME') if 'EXPTIME' in first_hdr else 1.0
        sub = dict(self.sub) if self.sub is not None else None
        os.makedirs(self.args.red_dir, exist_ok=True)
        t0 = time.time()
        live = {}                                                  # idx -> (fn, header, fits_out)
        stage, written, input_failed = None, {}, set()
        import threading
        all_written = threading.Event()
        kw = {}
        # threads and frames in flight from the cores this process may use (cgroup quota / affinity mask), like bench.py's
        # files-to-files run: on the 16 cores of a one-GPU box 6 lanes, 16 frames, 12 writers, 4 readers (one process), or 4 lanes,
        # 12 frames, 8 writers, 2 readers in each of two (list_processes)
        from blackbox_amd import pipeline as _pl
        cores = _pl.cpu_budget()
        lanes = max(2, min(6, cores // 2))
        depth = max(2, min(16 if cores >= 12 else (12 if cores >= 8 else 8), len(todo)))
        nwriters = max(2, min(12, cores))                     # (they wait in copies and write calls: 4 per 8 cores held the RAM-disk run at 70-75 frames/s, 8 give 90-95)
        nreaders = max(2, min(4, cores // 4))
        # (tuning runs: BBX_LIST_LANES / BBX_LIST_DEPTH / BBX_LIST_WRITERS / BBX_LIST_READERS override the choice above)
        lanes = int(os.environ.get('BBX_LIST_LANES', lanes))
        depth = max(2, min(int(os.environ.get('BBX_LIST_DEPTH', depth)), len(todo)))
        nwriters = int(os.environ.get('BBX_LIST_WRITERS', nwriters))
        nreaders = int(os.environ.get('BBX_LIST_READERS', nreaders))
        if self.args.fpack:
            from blackbox_amd import outstage
            ny, nx = 2 * geom.ysize_chan, 8 * geom.xsize_chan
            stage = outstage.OutputStage(self.ctx.device, ny, nx, nwriters=nwriters)

            def header_hook(f, hdrs):
                """the frame's scalars are in: complete the headers (bookkeeping, QC flags, subtraction keywords) through
                the very code of the serial path; its image writes only hand their headers to the stage"""
                fn, header, fits_out = live[f.idx]
                self._staged = dict(names=set(f.out_names.values()), headers={}, wanted=set(), deferred=[])
                try:
                    self._finish_from_pipeline(f, fn, header, fits_out, t0)
                    f.staged_wanted = set(self._staged['wanted'])
                    res = dict(hdrs)
                    res.update(self._staged['headers'])
                    jobs = self._staged['deferred']
                    if jobs:
                        # the frame's small files: one task of the host pool (a worker process formats and writes them);
                        # they count among the frame's files -- on_written fires when they and the images are on disk
                        g = f.out_group
                        with g.lock:
                            g.left += 1
                        from blackbox_amd import catalogs
                        pipe.pool.pool.apply_async(catalogs.write_small_products, (jobs,), callback=lambda r, g=g: g.file_done(None),
                                                   error_callback=lambda e, g=g: g.file_done(None, e))
                    return res
                finally:
                    self._staged = None

            def on_written(f, group):
                # products the frame's QC decided against (red flag: no subtraction products) were queued before the flag
                # was known: take them away again
                for p in group.paths:
                    if p not in getattr(f, 'staged_wanted', set(group.paths)):
                        try:
                            os.unlink(p)
                        except OSError:
                            pass
                written[f.idx] = group.error
                _FILES_DONE.append(time.time())
                if len(written) + len(input_failed) >= len(todo):
                    all_written.set()
            kw = dict(outstage=stage, out_base=lambda idx, h: todo[idx][1].replace('.fits', ''), on_written=on_written,
                      header_hook=header_hook, stage_limmag=self._limmag_is_flux)
        self._pipe_exptime = exptime
        pipe = FramePipeline(self.ctx, self.tel, geom, mflat=self.mflat, mbias=self.mbias, bpm=self.bpm,
                             xtalk_coeffs=self.xtalk, exptime=exptime, depth=depth, lanes=min(lanes, depth),
                             do_finish=True, detect_sats=True, keep_outputs=True, subtract=sub, log=log, **kw)

        from blackbox_amd import instage

        class _Serial:
            """read + upload a frame when the pipeline asks for the next one (raw frames that are not unsigned 16-bit); a file
            that cannot be read fails alone (instage.InputError in its place)"""

            def __init__(self_):
                self_.k = 0

            def __iter__(self_):
                return self_

            def __next__(self_):
                idx = self_.k
                if idx >= len(todo):
                    raise StopIteration
                self_.k += 1
                fn, fits_out = todo[idx]
                try:
                    d_raw, header = (first_raw, first_hdr) if idx == 0 else self.read_raw(fn)
                    if tuple(d_raw.shape) != (geom.ny_raw, geom.nx_raw):
                        raise ValueError('frames of different shapes in one --image_list: {} vs {}'.format(
                            tuple(d_raw.shape), (geom.ny_raw, geom.nx_raw)))
                except Exception as e:
                    raise instage.InputError(idx, fn, e)
                live[idx] = (fn, header, fits_out)
                return d_raw, header

        # unsigned 16-bit raw frames (plain or fpacked: what the telescopes deliver) come through the input stage: reader
        # threads, one upload per file, the Rice decode on the device, nothing of it on the orchestrating thread
        src = None
        if first_raw.dtype == self.torch.uint16 and len(todo) > 1:
            del first_raw
            src = instage.InputStage(self.ctx, [fn for fn, _ in todo], (geom.ny_raw, geom.nx_raw), nreaders=nreaders,
                                     nbuf=pipe.depth + 4, ahead=max(2, min(4, nreaders)))

            class _Source:
                """the input stage's frames with the bookkeeping this run keeps per frame"""
                has_next = staticmethod(src.has_next)

                def __iter__(self_):
                    return self_

                def __next__(self_):
                    raw, header, ev = next(src)
                    idx = src.taken - 1
                    live[idx] = (todo[idx][0], dict(header), todo[idx][1])
                    return raw, live[idx][1], ev

        def on_done(idx, f):
            if src is not None:
                src.release(f.raw)
            fn, header, fits_out = live[idx]
            try:
                if stage is None:
                    self._finish_from_pipeline(f, fn, header, fits_out, t0)
                out[fn] = getattr(f, 'written_name', None)
            except Exception:
                log.exception('exception was raised during [blackbox_reduce] of %s', fn)
                out[fn] = None
            if stage is None:
                live.pop(idx, None)
                _FILES_DONE.append(time.time())

        def on_input_error(idx, e):
            # this file fails (blackbox.py:948-999: exception logged, None for the file), the list goes on
            log.error('exception was raised while reading %s: %r', todo[idx][0], e.cause)
            out[todo[idx][0]] = None
            input_failed.add(idx)
            if stage is not None and len(written) + len(input_failed) >= len(todo):
                all_written.set()
        try:
            try:
                pipe.run(_Source() if src is not None else _Serial(), on_done=on_done, on_input_error=on_input_error)
            except Exception:
                # the pipeline itself gave up (not a file's fault): what it finished stands, the rest goes one by one below
                log.exception('pipelined run failed; the files without products are reduced one by one')
                for idx, (fn, fits_out) in enumerate(todo):
                    if fn not in out and idx not in written:
                        input_failed.add(idx)                        # (no product of the stage will come for it)
                all_written.set()
            if stage is not None and not all_written.wait(600.0):
                log.error('output stage: %d of %d frames written', len(written), len(todo))
            for idx, err in written.items():
                if err is not None:
                    log.error('writing the products of %s failed: %r', todo[idx][0], err)
                    out[todo[idx][0]] = None
        finally:
            n = max(1, pipe.t_stats[3])
            _PIPE_STATS.update(frames=pipe.t_stats[3], lanes=len(pipe.lane_thread), depth=pipe.depth,
                               ms_start_to_strip_statistics=round(1e3 * pipe.t_stats[0] / n, 2), ms_overscan_fits=round(1e3 * pipe.t_stats[1] / n, 2),
                               ms_device_stage_and_lane_queue=round(1e3 * pipe.t_stats[2] / n, 2))
            pipe.close()
            if stage is not None:
                stage.close()
            if src is not None:
                src.close()
        for fn, _ in todo:
            if fn not in out:
                out[fn] = self.reduce_logged(fn)
        return [out.get(fn) for fn in files]

    def _finish_from_pipeline(self, f, fn, header, fits_out, t0):
        """header completion + products of a frame that came out of the pipeline"""
        R = self.R
        # the pipeline works with one exposure time; NCOSMICS follows the frame's own
        et = R.hval(header, 'EXPTIME') if 'EXPTIME' in header else 1.0
        if et != f_exptime(self, f) and not isinstance(R.hval(header, 'NCOSMICS'), str):
            header['NCOSMICS'] = (R.hval(header, 'NCOSMICS') * float(f_exptime(self, f)) / float(et), header['NCOSMICS'][1])
        if 'zogy' in f.failed:
            header['Z-P'] = (False, 'successfully processed by ZOGY?')
        fh = self.open_image_log(fits_out)                    # the per-image log (blackbox.py:1311-1318) of a pipelined frame:
        try:                                                  # what its completion has to say (QC flags, failed steps)
            for step in f.failed:
                log.error('step [%s] failed for %s', step, fn)
            f.written_name = self.finish_object(fn, f.data, f.mask, header, f.hm, fits_out, t0, sub_result=f.sub)
        finally:
            self.close_image_log(fh)

    def read_header(self, filename):
        """the header of a raw frame without its pixels"""
        if filename.endswith('.fz'):
            h = self.fitsio.read_hdus(filename, headers_only=True)[1][0]
        else:
            h = self.fitsio.read_hdus(filename, headers_only=True)[0][0]
        header = dict(h)
        for k in ('BZERO', 'BSCALE', 'BITPIX', 'NAXIS', 'NAXIS1', 'NAXIS2', 'SIMPLE', 'EXTEND'):
            header.pop(k, None)
        return header


def f_exptime(reducer, f):
    """the exposure time the pipeline of this run works with"""
    return getattr(reducer, '_pipe_exptime', 1.0)


def list_processes(args, nfiles):
    """how many pipelined processes a list run uses on its GPU.  One interpreter drives the GPU through some 300 library
    calls and a dozen file writes per frame from ~25 threads that share its lock: at ~60-75 frames/s that lock, not the
    GPU, is what a 16-core host runs out of.  Two processes with half the list, half the cores and a GPU context each have
    two locks (the GPU time-slices their queues; the masters are in HBM twice)."""
    if not args.image_list or nfiles < 2 or args.nproc > 1:
        return 1
    nlp = args.list_procs or int(os.environ.get('BBX_LIST_PROCS', '0'))     # (the variable: tuning runs)
    if nlp > 0:
        return min(nlp, nfiles)
    from blackbox_amd import pipeline as _pl
    world = int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1)
    return 2 if (_pl.cpu_budget() // max(1, world) >= 16 and nfiles >= 32) else 1


def reduce_list_in_processes(argv, files, nproc):
    """--image_list over [nproc] child processes `python blackbox.py ... --image_list <share> --list_procs 1` (the file k of
    the list goes to child k mod nproc; each child is told its share of the cores: BBX_CPU_BUDGET).  The parent makes no
    GPU call.  -> the results in the order of [files]; a child that dies reports None for the files it had not finished"""
    import json
    import subprocess
    import tempfile
    from blackbox_amd import pipeline as _pl
    cores = _pl.cpu_budget()
    base = []
    skip = False
    for a in argv:                                               # the parent's flags without the list and the split
        if skip:
            skip = False
            continue
        if a in ('--image_list', '--list_procs', '--image'):
            skip = True
            continue
        if a.startswith('--image_list=') or a.startswith('--list_procs=') or a.startswith('--image='):
            continue
        base.append(a)
    td = tempfile.mkdtemp(prefix='bbx_list_')
    procs = []
    try:
        for k in range(nproc):
            share = files[k::nproc]
            lst = os.path.join(td, 'share_%d.txt' % k)
            with open(lst, 'w') as f:
                f.write('\n'.join(share) + '\n')
            env = dict(os.environ, BBX_CPU_BUDGET=str(max(2, cores // nproc)), BBX_TIMING='1')
            procs.append((share, subprocess.Popen([sys.executable, os.path.abspath(__file__)] + base + ['--image_list', lst, '--list_procs', '1'],
                                                  env=env, stdout=subprocess.PIPE, text=True)))
        _mark('list_processes_started')
        out = {}
        done, stats = [], []
        for share, p in procs:
            text, _ = p.communicate()
            lines = [ln for ln in text.splitlines() if ln.strip()]
            tm = [ln for ln in lines if ln.startswith('BBX_TIMING ')]
            res = [ln for ln in lines if not ln.startswith('BBX_TIMING ')][-len(share):]
            if p.returncode != 0 or len(res) != len(share):
                log.error('list process for %d files ended with code %s and %d result lines', len(share), p.returncode, len(res))
                res = (res + ['None'] * len(share))[:len(share)] if p.returncode == 0 else ['None'] * len(share)
            for fn, r in zip(share, res):
                out[fn] = None if r == 'None' else r
            if tm:
                t = json.loads(tm[-1][len('BBX_TIMING '):])
                done += t.get('files_done_unix', [])
                stats.append(dict(pipeline=t.get('pipeline'), hbm_peak_GB_tensors=t.get('hbm_peak_GB_tensors'), marks=t.get('marks'), t0=t.get('t_module_import_unix', _T0)))
    finally:
        for _, p in procs:
            if p.poll() is None:
                p.kill()
        import shutil
        shutil.rmtree(td, ignore_errors=True)
    res = [out.get(fn) for fn in files]
    for o in res:
        print(o)
    if os.environ.get('BBX_TIMING'):
        _mark('done')
        # (the moment the last child had its masters in HBM, on this process's clock)
        first = [s_['t0'] + dict(s_['marks']).get('calibration_and_reference_files_in_hbm', 0.0) - _T0 for s_ in stats if s_.get('marks')]
        print('BBX_TIMING ' + json.dumps(dict(marks=_MARKS + [('calibration_and_reference_files_in_hbm', max(first) if first else 0.0)],
                                              t_module_import_unix=_T0, files_done_unix=sorted(done), list_processes=nproc,
                                              pipeline=[s_['pipeline'] for s_ in stats],
                                              hbm_peak_GB_tensors=round(sum(s_.get('hbm_peak_GB_tensors') or 0.0 for s_ in stats), 2))))
    return res


def build_parser():
    ap = argparse.ArgumentParser(description='BlackBOX per-image reduction on MI355X')
    ap.add_argument('--telescope', type=str, default='ML1')
    ap.add_argument('--mode', type=str, default='day')
    ap.add_argument('--date', type=str, default=None)
    ap.add_argument('--read_path', type=str, default=None)
    ap.add_argument('--recursive', type=str2bool, default=False)
    ap.add_argument('--imgtypes', type=str, default=None)
    ap.add_argument('--filters', type=str, default=None)
    ap.add_argument('--image', type=str, default=None)
    ap.add_argument('--image_list', type=str, default=None)
    ap.add_argument('--img_reduce', type=str2bool, default=True)
    ap.add_argument('--cat_extract', type=str2bool, default=False)
    ap.add_argument('--trans_extract', type=str2bool, default=False)
    ap.add_argument('--force_reproc_new', type=str2bool, default=False)
    ap.add_argument('--master_date', type=str, default=None)
    ap.add_argument('--name_genlog', type=str, default=None)
    ap.add_argument('--keep_tmp', type=str2bool, default=None)
    # explicit calibration inputs (the date-based master selection of master_prep is orchestration)
    ap.add_argument('--mflat', type=str, default=None)
    ap.add_argument('--fpack', type=lambda v: str(v).lower() in ('1', 'true', 'yes'), default=False,
                    help='write tile-compressed .fits.fz products (compressed on the GPU)')
    ap.add_argument('--mbias', type=str, default=None)
    ap.add_argument('--bpm', type=str, default=None)
    ap.add_argument('--crosstalk', type=str, default=None)
    ap.add_argument('--nonlin', type=str, default=None, help='pickle of 16 scipy splines (set_bb.nonlin_corr_file)')
    ap.add_argument('--red_dir', type=str, default=None)
    ap.add_argument('--ysize_chan', type=int, default=None)
    ap.add_argument('--xsize_chan', type=int, default=None)
    # explicit inputs of zogy.optimal_subtraction (reference selection / PSFEx are orchestration)
    ap.add_argument('--ref', type=str, default=None, help='reference image on the new frame\'s pixel grid (_red.fits)')
    ap.add_argument('--ref_mask', type=str, default=None)
    ap.add_argument('--ref_bkg_std_mini', type=str, default=None, help='the reference\'s _bkg_std_mini.fits')
    ap.add_argument('--ref_bkgsub', type=str2bool, default=False, help='reference is background-subtracted (BKG-SUB)')
    ap.add_argument('--psf_new', type=str, default=None, help='PSFEx .psf model or FITS stamp / cube of the new image')
    ap.add_argument('--psf_ref', type=str, default=None)
    ap.add_argument('--fratio', type=float, default=1.0, help='flux ratio new / ref (Z-FNR)')
    ap.add_argument('--zogy_dx', type=float, default=0.0, help='[pix] astrometric scatter in x (Z-DXSTD)')
    ap.add_argument('--zogy_dy', type=float, default=0.0)
    ap.add_argument('--subimage_size', type=int, default=None)
    ap.add_argument('--subimage_border', type=int, default=None)
    ap.add_argument('--bkg_boxsize', type=int, default=None)
    ap.add_argument('--zeropoint', type=float, default=None,
                    help='[mag] photometric zeropoint for 1 e-/s (else header PC-ZP): _trans_limmag in magnitudes')
    ap.add_argument('--nproc', type=int, default=1, help='worker processes for --image_list (one GPU context each)')
    ap.add_argument('--list_procs', type=int, default=0,
                    help='--image_list: this many pipelined processes share the GPU, each with a share of the list and of the cores '
                         '(0 = by the cores this process may use; 1 = one process)')
    return ap


def main(argv=None):
    ap = build_parser()
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)
    logging.basicConfig(level='INFO', format='%(asctime)s [%(levelname)s, %(process)s] %(message)s')
    for flag in ('date', 'read_path', 'imgtypes', 'filters', 'master_date', 'name_genlog'):
        if getattr(args, flag):
            log.warning('--%s concerns orchestration and is ignored by the hot-path build', flag)
    if args.mode != 'day':
        log.warning('night mode (watchdog) is out of scope; running the given files once')
    files = []
    if args.image:
        files.append(args.image)
    if args.image_list:
        with open(args.image_list) as f:
            files += [ln.strip() for ln in f if ln.strip()]
    if not files:
        ap.error('--image or --image_list required')
    tel = telescope_of(args, files[0])
    from blackbox_amd import farm
    mine = farm.shard(files)
    if args.nproc > 1 and len(mine) > 1:
        # the reference's farm (blackbox.py:375-379): a pool of workers over the file list, one GPU context each
        configure(argv)
        _mark('pool_start')
        try:
            out = pool_func(try_blackbox_reduce, mine, nproc=args.nproc)
        except WrapException as e:
            log.error('a worker raised during [blackbox_reduce]:\n%s', e.formatted)
            raise
        for o in out:
            print(o)
        if os.environ.get('BBX_TIMING'):
            import json
            _mark('done')
            print('BBX_TIMING ' + json.dumps(dict(marks=_MARKS, t_module_import_unix=_T0)))
        return out
    nlp = list_processes(args, len(mine))
    if nlp > 1:
        return reduce_list_in_processes(argv, mine, nlp)
    red = Reducer(tel, args)
    sampler = _start_sampler(os.environ.get('BBX_CLI_SAMPLE'))
    if args.image_list and len(mine) > 1:
        out = red.reduce_list(mine)
    else:
        out = [red.reduce_logged(f) for f in mine]
    for o in out:
        print(o)
    if os.environ.get('BBX_TIMING'):
        import json
        _mark('done')
        import torch
        print('BBX_TIMING ' + json.dumps(dict(marks=_MARKS, t_module_import_unix=_T0, files_done_unix=sorted(_FILES_DONE), pipeline=_PIPE_STATS,
                                              hbm_peak_GB_tensors=round(torch.cuda.max_memory_allocated() / 1e9, 2))))
    return out


if __name__ == '__main__':
    main()
    # Every product is on disk and every line printed: leave without the interpreter's teardown (module destructors, the
    # GPU runtime's unload, the worker pools' joins: 0.3-0.6 s of a 2 s per-file process -- the reference's Slurm contract
    # is one process per file).  An exception never gets here: it ends the process the ordinary way, with its traceback.
    logging.shutdown()
    sys.stdout.flush()
    sys.stderr.flush()
    if os.environ.get('BBX_FAST_EXIT', '1') != '0':
        os._exit(0)
