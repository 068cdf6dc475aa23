#!/usr/bin/env python
"""blackbox.py -- per-file entry point of the MI355X reduction, same command line as the
reference's `python blackbox.py --telescope T --img_reduce True ... --image FILE`
(blackbox.py:8128-8213, blackbox_slurm_google.py:305-309).

FITS in -> `{tel}_{yyyymmdd}_{hhmmss}_red.fits` (float32, e-) + `..._mask.fits` (uint8) out.
Flags of the reference that concern orchestration (night mode, dates, master creation,
catalogues) are accepted and ignored with a warning: only the per-image reduction hot
path lives here.  With WORLD_SIZE > 1 (torch.distributed.run) every rank takes every
world-th file of --image_list on its own GPU.
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


def str2bool(v):
    """blackbox.py:8115-8123"""
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def outname(header, tel, red_dir):
    """{tel}_{yyyymmdd}_{hhmmss}_red.fits from DATE-OBS (blackbox.py:1004-1022, 1166-1184)"""
    from blackbox_amd.reduce import hval
    date_obs = str(hval(header, 'DATE-OBS')) if 'DATE-OBS' in header else time.strftime('%Y-%m-%dT%H:%M:%S')
    d, t = date_obs.split('T')[0].replace('-', ''), date_obs.split('T')[-1].split('.')[0].replace(':', '')
    return os.path.join(red_dir, '{}_{}_{}_red.fits'.format(tel, d, t))


class Reducer:
    """per-process state: GPU context + masters resident in HBM"""

    def __init__(self, tel, args):
        import torch
        from blackbox_amd import fitsio, reduce as R
        self.R, self.torch, self.fitsio = R, torch, fitsio
        _, _, local = __import__('blackbox_amd.farm', fromlist=['rank_world']).rank_world()
        self.ctx = R.Context(local)
        self.tel = tel
        dev = self.ctx.device

        def load(path, dtype):
            if not path:
                return None
            return torch.from_numpy(np.ascontiguousarray(fitsio.read_image(path, dtype=dtype))).to(dev)
        self.mflat = load(args.mflat, np.float32)
        self.mbias = load(args.mbias, np.float32)
        self.bpm = load(args.bpm, np.uint8)
        if self.bpm is not None and bool((self.bpm & 12).any()):
            raise ValueError('bad-pixel mask carries saturated(4)/saturated-connected(8) bits; expected 1 and 32 only')
        self.xtalk = R.read_crosstalk(args.crosstalk) if args.crosstalk else None
        self.args = args

    def try_blackbox_reduce(self, filename):
        """blackbox.py:948-999: returns the reduced file name or None; never raises"""
        try:
            return self.blackbox_reduce(filename)
        except Exception:
            log.exception('exception was raised during [blackbox_reduce] of %s', filename)
            return None

    def blackbox_reduce(self, filename):
        R, torch, fitsio = self.R, self.torch, self.fitsio
        t0 = time.time()
        d_raw = None
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
        header = dict(hraw)
        for k in ('BZERO', 'BSCALE', 'BITPIX', 'NAXIS', 'NAXIS1', 'NAXIS2', 'SIMPLE', 'EXTEND'):
            header.pop(k, None)
        red_dir = self.args.red_dir or os.path.dirname(os.path.abspath(filename))
        fits_out = outname(header, self.tel, red_dir)
        if os.path.isfile(fits_out) and os.path.isfile(fits_out.replace('_red', '_mask')) \
                and not (self.args.img_reduce and self.args.force_reproc_new):
            log.info('%s already reduced; skipping', filename)           # blackbox.py:1336-1390
            return fits_out
        if d_raw is None:
            d_raw = torch.from_numpy(np.ascontiguousarray(raw)).to(self.ctx.device)
        exptime = R.hval(header, 'EXPTIME') if 'EXPTIME' in header else 1.0
        imgtype = str(R.hval(header, 'IMAGETYP')).lower() if 'IMAGETYP' in header else 'object'
        if imgtype == 'flat':
            # flat frames (blackbox.py:1826-1850): overscan, master bias, mask; no flat division,
            # cosmics, crosstalk or trails; header statistics from get_flatstats
            from blackbox_amd import flatstats
            data, mask, header, hm = R.reduce_object(
                self.ctx, d_raw, header, self.tel, mflat=None, mbias=self.mbias, bpm=self.bpm, exptime=exptime,
                ysize_chan=self.args.ysize_chan, xsize_chan=self.args.xsize_chan, do_cosmics=False, detect_sats=False)
            flatstats.get_flatstats(self.ctx, data, header, mask, self.tel, ysize_chan=self.args.ysize_chan,
                                    xsize_chan=self.args.xsize_chan)
        else:
            data, mask, header, hm = R.reduce_object(
                self.ctx, d_raw, header, self.tel, mflat=self.mflat, mbias=self.mbias, bpm=self.bpm,
                xtalk_coeffs=self.xtalk, exptime=exptime, ysize_chan=self.args.ysize_chan,
                xsize_chan=self.args.xsize_chan)
        header['BUNIT'] = ('e-', 'pixel values are in electrons')
        # bookkeeping keywords of the header contract (blackbox.py:1110-1112, 1520, 1607, 1669-1690,
        # 1817-1855; verify_header 2893-3255)
        from blackbox_amd import __version__ as bbx_version
        base = lambda p: os.path.basename(p) if p else 'None'                  # noqa: E731
        stem = lambda p: os.path.basename(p).split('.fits')[0] if p else 'None'  # noqa: E731
        header['BB-V'] = ('amd-' + bbx_version, 'BlackBOX version used')
        header['KW-V'] = (KEYWORDS_VERSION, 'header keywords version used')
        header['BB-START'] = (time.strftime('%Y-%m-%dT%H:%M:%S', time.gmtime(t0)), 'start UTC date of BlackBOX image run')
        header['XTALK-F'] = (base(self.args.crosstalk), 'name crosstalk coefficients file')
        header['NONLIN-F'] = ('None', 'name non-linearity correction file')
        header['MBIAS-F'] = (stem(self.args.mbias) if R.hval(header, 'MBIAS-P') else 'None', 'name of master bias applied')
        header['MFLAT-F'] = (stem(self.args.mflat) if R.hval(header, 'MFLAT-P') else 'None', 'name of master flat applied')
        for k, c in (('XTALK-P', 'corrected for crosstalk?'), ('COSMIC-P', 'corrected for cosmic rays?'),
                     ('SAT-P', 'processed for satellite trails?')):
            header.setdefault(k, (False, c))                                     # steps that did not run
        header.setdefault('NCOSMICS', ('None', '[/s] number of cosmic rays identified'))
        header.setdefault('NSATS', ('None', 'number of satellite trails identified'))
        header['MFRING-P'] = (False, 'corrected for master fringe map?')
        header['MFRING-F'] = ('None', 'name of master fringe map applied')
        header['FRRATIO'] = ('None', 'fringe ratio (science/fringe map) applied')
        # QC flags on the reduction keywords (blackbox.py:2000; qc.py)
        from blackbox_amd import qc
        qc_flag = qc.run_qc_check(header, self.tel, check_key_type='full')      # flags go into the image header
        if qc_flag == 'red':
            log.error('red QC flag for %s', filename)
        redfile = os.path.basename(fits_out).split('.fits')[0]
        header['REDFILE'] = (redfile, 'BlackBOX reduced image name')
        header['MASKFILE'] = (redfile.replace('_red', '_mask'), 'BlackBOX mask image name')
        qc.verify_header(header, ['full'], name=fits_out)                     # blackbox.py:2062 (raises on a DB keyword)
        os.makedirs(red_dir, exist_ok=True)
        if self.args.fpack:
            # products leave the GPU tile-compressed (reference: fpack of the files kept,
            # copy_files2keep 4033-4035): only the compressed bytes cross PCIe
            from blackbox_amd import fpack as P
            fits_out = P.fpack_image(self.ctx, fits_out, data, header)
            P.fpack_image(self.ctx, fits_out.replace('_red', '_mask'), mask, hm)
        else:
            fitsio.write_image(fits_out, data.cpu().numpy(), header)
            fitsio.write_image(fits_out.replace('_red', '_mask'), mask.cpu().numpy(), hm)
        log.info('reduced %s -> %s in %.2f s', filename, fits_out, time.time() - t0)
        return fits_out


def main(argv=None):
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
    ap.add_argument('--red_dir', type=str, default=None)
    ap.add_argument('--ysize_chan', type=int, default=None)
    ap.add_argument('--xsize_chan', type=int, default=None)
    args = ap.parse_args(argv)
    logging.basicConfig(level='INFO', format='%(asctime)s [%(levelname)s, %(process)s] %(message)s')
    for flag in ('date', 'read_path', 'imgtypes', 'filters', 'master_date', 'name_genlog'):
        if getattr(args, flag):
            log.warning('--%s concerns orchestration and is ignored by the hot-path build', flag)
    if args.mode != 'day':
        log.warning('night mode (watchdog) is out of scope; running the given files once')
    if args.cat_extract or args.trans_extract:
        log.warning('catalogue / transient extraction (ZOGY) is not part of this round; reduction only')
    files = []
    if args.image:
        files.append(args.image)
    if args.image_list:
        with open(args.image_list) as f:
            files += [ln.strip() for ln in f if ln.strip()]
    if not files:
        ap.error('--image or --image_list required')
    tel = args.telescope
    base = os.path.basename(files[0])
    for t in ('ML1', 'BG2', 'BG3', 'BG4'):                       # telescope from the file name (blackbox.py:145-152)
        if base.startswith(t):
            tel = t
    from blackbox_amd import farm
    mine = farm.shard(files)
    red = Reducer(tel, args)
    out = [red.try_blackbox_reduce(f) for f in mine]
    for o in out:
        print(o)
    return out


if __name__ == '__main__':
    main()
