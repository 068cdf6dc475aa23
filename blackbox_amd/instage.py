"""Input stage: raw frames come off the disk into HBM without stalling the pipeline that reduces them.

The reference reads a raw frame with `read_hdulist(filename, dtype='float32')` (blackbox.py:1451; astropy decompresses an
fpacked `.fits.fz` on the host).  Here a pool of reader threads does the host part -- file bytes into a pinned buffer,
the few header cards, the tile descriptors -- and leaves the pixels to the device:

    reader thread : file -> pinned buffer (readinto) -> header + descriptor table parsed on the host
                    -> ONE host-to-device copy of [compressed heap | descriptors] on the thread's own stream
                    -> bbx_funpack_tiles (Rice decode, uint16 raws with BZERO 32768) into a raw buffer of the pool
                    -> event
    consumer      : iterates (raw device tensor, header, event) in file order; FramePipeline makes its first stream wait
                    for the event, nothing on the host waits for the GPU

Uncompressed `.fits` raws go the same way without the decode (the pixels are byte-swapped on the device).
Raw buffers come from a pool of [nbuf] tensors and return to it through release() when the frame is done.
Everything a reader thread waits on releases the GIL (file reads, stream synchronisation).
"""
import ctypes as C
import queue
import threading

import numpy as np
import torch

from . import fitsio, fpack
from ._lib import lib, check, wait_event


class InputError(Exception):
    """one file of the list could not be read or decoded (blackbox.py:948-999: the reference fails that file and goes on):
    raised by the iterator in that file's place; the frames behind it follow as if nothing had happened"""

    def __init__(self, idx, path, cause):
        Exception.__init__(self, '{}: {!r}'.format(path, cause))
        self.idx, self.path, self.cause = idx, path, cause


class RawFile:
    """what the host needs from a raw frame file: header dict, geometry, where the pixels (or their tile streams) are"""
    __slots__ = ('header', 'ny', 'nx', 'compressed', 'data_off', 'data_len', 'desc', 'heap_off', 'heap_len', 'bitpix', 'bzero', 'bytepix')


def parse_raw(buf, nbytes):
    """buf: the file's bytes (numpy uint8 view).  -> RawFile.  Supported: a primary or IMAGE HDU of BITPIX 16 (the
    raw frames: unsigned with BZERO 32768) / -32, or a RICE_1 tile-compressed image of those types tiled by rows (fpack's
    default) -- what `fpack` leaves of a raw frame."""
    hv = fitsio._hv
    pos = 0
    out = None
    while pos < nbytes:
        h, hlen = _header_at(buf, pos, nbytes)
        pos += hlen
        naxis = int(hv(h, 'NAXIS', 0) or 0)
        bitpix = int(hv(h, 'BITPIX', 8))
        if naxis == 0:
            continue
        shape = [int(hv(h, 'NAXIS%d' % k)) for k in range(naxis, 0, -1)]
        dbytes = int(np.prod(shape)) * abs(bitpix) // 8
        pcount = int(hv(h, 'PCOUNT', 0) or 0)
        r = RawFile()
        r.header = h
        if hv(h, 'ZIMAGE', False) is True:
            if str(hv(h, 'ZCMPTYPE')).strip() not in ('RICE_1', 'RICE_ONE'):
                raise ValueError('compression {} not supported (RICE_1 only)'.format(hv(h, 'ZCMPTYPE')))
            r.ny, r.nx, r.bitpix = int(hv(h, 'ZNAXIS2')), int(hv(h, 'ZNAXIS1')), int(hv(h, 'ZBITPIX'))
            if int(hv(h, 'ZNAXIS')) != 2 or int(hv(h, 'ZTILE1', r.nx)) != r.nx or int(hv(h, 'ZTILE2', 1)) != 1:
                raise ValueError('only 2-D images tiled by rows are supported')
            zpar = {str(hv(h, 'ZNAME%d' % k)).strip(): hv(h, 'ZVAL%d' % k) for k in range(1, 5) if ('ZNAME%d' % k) in h}
            if int(zpar.get('BLOCKSIZE', 32)) != 32:
                raise ValueError('Rice block size {} not supported'.format(zpar.get('BLOCKSIZE')))
            r.bytepix = int(zpar.get('BYTEPIX', 4))
            if r.bitpix not in (16, 32) or r.bytepix != r.bitpix // 8:
                raise ValueError('tile-compressed raw frames: integer images expected (ZBITPIX {})'.format(r.bitpix))
            lay = fpack._table_layout(h)
            rowlen = shape[1]
            o = lay['COMPRESSED_DATA'][0]
            if pos + r.ny * rowlen > nbytes:
                raise EOFError('truncated file: the tile table ends at byte {}, {} bytes read'.format(pos + r.ny * rowlen, nbytes))
            tb = buf[pos:pos + r.ny * rowlen].reshape(r.ny, rowlen)
            r.desc = np.ascontiguousarray(tb[:, o:o + 8]).view('>i4').astype(np.int32)
            r.compressed, r.heap_off, r.heap_len = True, pos + dbytes, pcount
            if (r.desc < 0).any() or int((r.desc[:, 0].astype(np.int64) + r.desc[:, 1]).max(initial=0)) > pcount:
                raise ValueError('tile descriptors point outside the heap')
            if r.heap_off + r.heap_len > nbytes:
                # (a file still being written, a broken transfer: what was read so far parses, its tail would be decoded from
                # whatever the buffer held before -- astropy raises on such a file, so does this)
                raise EOFError('truncated file: the heap ends at byte {}, {} bytes read'.format(r.heap_off + r.heap_len, nbytes))
            r.bzero = hv(h, 'BZERO', 0)
            out = r
        elif naxis == 2 and str(hv(h, 'XTENSION', 'IMAGE')).strip() in ('IMAGE',):
            r.ny, r.nx, r.bitpix = shape[0], shape[1], bitpix
            r.compressed, r.data_off, r.data_len = False, pos, dbytes
            if pos + dbytes > nbytes:
                raise EOFError('truncated file: the image ends at byte {}, {} bytes read'.format(pos + dbytes, nbytes))
            r.bzero = hv(h, 'BZERO', 0)
            out = r
        nb = dbytes + pcount
        pos += nb + (-nb) % fitsio.BLOCK
    if out is None:
        raise ValueError('no image in the file')
    skip = ('ZIMAGE', 'ZCMPTYPE', 'ZBITPIX', 'ZNAXIS', 'ZTILE', 'ZNAME', 'ZVAL', 'ZQUANTIZ', 'ZDITHER0', 'ZTENSION', 'ZPCOUNT',
            'ZGCOUNT', 'TTYPE', 'TFORM', 'TFIELDS', 'XTENSION', 'PCOUNT', 'GCOUNT', 'NAXIS', 'BITPIX', 'EXTNAME', 'BZERO', 'BSCALE',
            'SIMPLE', 'EXTEND')
    out.header = {k: v for k, v in out.header.items() if not any(k.startswith(p) for p in skip)}
    return out


def _header_at(buf, pos, nbytes):
    """the header unit that starts at byte [pos] -> (dict, its length in bytes)"""
    import io
    end = pos
    while True:
        if end + fitsio.BLOCK > nbytes:
            raise EOFError('truncated FITS header')
        block = bytes(buf[end:end + fitsio.BLOCK])
        end += fitsio.BLOCK
        if any(block[i:i + 8] == b'END     ' for i in range(0, fitsio.BLOCK, 80)):
            break
    return fitsio._read_header(io.BytesIO(bytes(buf[pos:end]))), end - pos


class InputStage:
    """reader pool + raw-buffer pool.  Iterate it for (raw, header, event) in the order of [files]; call release(raw) when a
    frame is done with its raw buffer.  files: list of paths, or a callable(idx) -> path | None for an endless source."""

    def __init__(self, ctx, files, shape, nreaders=3, nbuf=8, ahead=4, max_file_bytes=None):
        self.ctx, self.device = ctx, ctx.device
        self.ny, self.nx = shape
        self.files = files
        self.nfiles = None if callable(files) else len(files)
        self.pool = queue.Queue()
        self.bufs = [torch.empty(shape, dtype=torch.uint16, device=self.device) for _ in range(nbuf)]
        for b in self.bufs:
            self.pool.put(b)
        self.rnd = None
        self.cap = int(max_file_bytes or (self.ny * self.nx * 2 + 4 * self.ny * 8 + 64 * fitsio.BLOCK))
        self.ahead = threading.Semaphore(ahead)                 # frames decoded ahead of the consumer
        self.next_idx, self.idx_lock, self.taken = 0, threading.Lock(), 0
        self.ready, self.cv = {}, threading.Condition()
        self.stop = False
        self.bytes_read = 0
        self.threads = [threading.Thread(target=self._reader, args=(k,), daemon=True, name='bbx-reader') for k in range(nreaders)]
        for t in self.threads:
            t.start()

    # ---- consumer side ---------------------------------------------------------------------------------
    def __iter__(self):
        return self

    def has_next(self):
        """True when __next__ would not block (the next frame is decoded, or the source has ended); FramePipeline asks before
        it takes a frame, so that its orchestrating thread never sleeps inside the iterator"""
        with self.cv:
            return self.stop or self.taken in self.ready or (self.nfiles is not None and self.taken >= self.nfiles)

    def __next__(self):
        k = self.taken
        if self.nfiles is not None and k >= self.nfiles:
            raise StopIteration
        with self.cv:
            while k not in self.ready:
                if self.stop:
                    raise StopIteration
                self.cv.wait(0.5)
            item = self.ready.pop(k)
        self.taken = k + 1
        self.ahead.release()
        if item is None:
            self.nfiles = k                                      # the source ended
            raise StopIteration
        if isinstance(item, BaseException):
            raise InputError(k, self.files[k] if not callable(self.files) else None, item)
        return item

    def release(self, raw):
        self.pool.put(raw)

    def close(self):
        self.stop = True
        with self.cv:
            self.cv.notify_all()
        for _ in self.threads:
            self.ahead.release()
            self.pool.put(None)
        for t in self.threads:
            t.join(10.0)
        self.bufs = []

    # ---- reader thread ------------------------------------------------------------------------------------
    def _reader(self, k):
        torch.cuda.set_device(self.device)
        stream = torch.cuda.Stream(device=self.device)
        sp = C.c_void_p(stream.cuda_stream)
        pinned = torch.empty(self.cap + 16, dtype=torch.uint8, pin_memory=True)
        hbuf = pinned.numpy()
        staging = torch.empty(self.cap + 16, dtype=torch.uint8, device=self.device)
        d_desc = torch.empty(2 * self.ny, dtype=torch.int32, device=self.device)
        h_desc = torch.empty(2 * self.ny, dtype=torch.int32, pin_memory=True)
        done = torch.cuda.Event()                                # the staging buffers are free again
        while not self.stop:
            self.ahead.acquire()
            if self.stop:
                return
            with self.idx_lock:
                idx = self.next_idx
                self.next_idx += 1
            path = self.files(idx) if callable(self.files) else (self.files[idx] if idx < self.nfiles else None)
            if path is None:
                self._put(idx, None)
                return
            raw = self.pool.get()
            if raw is None:
                return
            queued = False                                        # device work of this frame is on the stream
            try:
                wait_event(done)                                 # this thread's previous frame has left the staging buffers
                with open(path, 'rb', buffering=0) as f:
                    n = 0
                    while True:
                        got = f.readinto(memoryview(hbuf)[n:self.cap])
                        if not got:
                            break
                        n += got
                if n >= self.cap:
                    raise ValueError('{}: larger than the staging buffer ({} bytes)'.format(path, self.cap))
                self.bytes_read += n
                r = parse_raw(hbuf, n)
                if (r.ny, r.nx) != (self.ny, self.nx):
                    raise ValueError('{}: frame of shape {} expected, got {}'.format(path, (self.ny, self.nx), (r.ny, r.nx)))
                with torch.cuda.stream(stream):
                    queued = True
                    if r.compressed:
                        if not (r.bitpix == 16 and r.bzero == 32768):
                            raise ValueError('{}: unsigned 16-bit raw frame expected'.format(path))
                        a = r.heap_off & ~15                     # (the copy starts 16-byte aligned; the kernel takes any 4-byte offset)
                        nb = r.heap_off + r.heap_len - a
                        hbuf[r.heap_off + r.heap_len:r.heap_off + r.heap_len + 16] = 0
                        staging[:nb + 16].copy_(pinned[a:a + nb + 16], non_blocking=True)
                        h_desc.numpy()[:] = r.desc.reshape(-1)
                        d_desc.copy_(h_desc, non_blocking=True)
                        check(lib.bbx_funpack_tiles(self.ctx.h, r.ny, r.nx, 2, C.c_void_p(d_desc.data_ptr()),
                                                    C.c_void_p(staging.data_ptr() + (r.heap_off - a)), 1, C.c_void_p(raw.data_ptr()),
                                                    None, None, 0, None, sp), 'bbx_funpack_tiles', self.ctx.h)
                    else:
                        if not (r.bitpix == 16 and r.bzero == 32768):
                            raise ValueError('{}: unsigned 16-bit raw frame expected'.format(path))
                        a = r.data_off & ~15
                        nb = r.data_off + r.data_len - a
                        staging[:nb].copy_(pinned[a:a + nb], non_blocking=True)
                        # big-endian int16 + BZERO -> uint16: bytes swapped, sign bit flipped, one pass (bbx_raw_be16)
                        check(lib.bbx_raw_be16(C.c_void_p(staging.data_ptr() + (r.data_off - a)), C.c_void_p(raw.data_ptr()), r.ny * r.nx, sp),
                              'bbx_raw_be16', self.ctx.h)
                    done.record(stream)
                    ev = torch.cuda.Event()
                    ev.record(stream)
                self._put(idx, (raw, r.header, ev))
            except BaseException as e:
                # this file fails, the thread and its buffers live on: whatever was queued for the frame leaves the staging
                # buffers and the raw buffer first, the raw buffer goes back to the pool, the consumer gets the exception
                # in the file's place (in order)
                try:
                    if queued:
                        stream.synchronize()
                except BaseException:
                    pass
                self.pool.put(raw)
                self._put(idx, e)
                if not isinstance(e, Exception):                  # KeyboardInterrupt / SystemExit: stop reading
                    return

    def _put(self, idx, item):
        with self.cv:
            self.ready[idx] = item
            self.cv.notify_all()
