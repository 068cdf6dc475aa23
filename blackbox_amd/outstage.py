"""Output stage: the products of a frame leave the GPU as tile-compressed FITS (`.fits.fz`) without
stalling the lane that made them.

The reference writes its products with astropy and fpacks the files it keeps afterwards
(blackbox.py:1981-1990 write, 812-857 fpack, 3933-4035 copy_files2keep).  Here the compression runs on
the device, queued on the lane's stream right behind the kernels that produced the image
(bbx_fpack_body: tile streams, offsets and the big-endian descriptor table in one enqueue); a pool of host
writer threads takes over from there:

    lane thread    : bbx_fpack_body -> async copy of the 4-number summary -> event          (no host wait)
    writer thread  : wait for the event -> device-to-host copy of exactly table + heap bytes into its pinned
                     buffer (own copy stream) -> device slot free again -> rows the quantiser refused: fetched and
                     gzip-compressed (CFITSIO's GZIP_COMPRESSED_DATA column) -> header cards -> file

Everything a writer thread waits on releases the GIL (event / stream synchronisation, zlib, file writes).
The files are byte-identical to what fpack.fpack_image writes.
"""
import ctypes as C
import gzip
import os
import queue
import threading
import time

import numpy as np
import torch

from . import fitsio, fpack
from ._lib import lib, check, wait_event

MAX_LIST = 4096                      # rows per image the quantiser may refuse before the image falls back to the serial path


def default_quant(path):
    """quantisation level by product name, as the reference fpacks them (blackbox.py:826-845)"""
    return 2 if ('Scorr' in path or 'limmag' in path) else (4 if 'Fpsf' in path else 16)


class _Slot:
    __slots__ = ('d_body', 'd_info', 'h_info', 'ev', 'cap')


class _Job:
    __slots__ = ('lane', 'slot', 'img', 'path', 'quant', 'seed', 'bitpix', 'shape', 'header', 'header_ready', 'group', 'bzero', 'scale')


class GroupCancelled(RuntimeError):
    """the frame this file belongs to was given up (its run failed or was closed): nothing is written"""


class _OnStream:
    """a library context seen through another stream (what fpack.fpack_image asks of a context: .h, .device, .stream())"""

    def __init__(self, ctx, stream):
        self.h, self.device, self._sp = ctx.h, ctx.device, C.c_void_p(stream.cuda_stream)

    def stream(self):
        return self._sp


class FrameGroup:
    """the files of one frame: done() fires when the last of them is on disk"""

    def __init__(self, token, on_done):
        self.token, self.on_done = token, on_done
        # left counts the files still to be written + 1 while the frame may still submit more (seal() takes that one)
        self.left, self.lock, self.paths, self.error = 1, threading.Lock(), [], None
        self.header_ready = threading.Event()
        self.headers = {}                      # path -> header dict; None -> the header of every other file
        self.cancelled = False

    def set_headers(self, headers):
        self.headers.update(headers)
        self.header_ready.set()

    def cancel(self, err=None):
        """the frame will never get its headers (its run failed): writers waiting for them skip the frame's files and
        release what they hold; the group reports [err]"""
        with self.lock:
            self.cancelled = True
            if self.error is None:
                self.error = err if err is not None else GroupCancelled('frame cancelled')
        self.header_ready.set()

    def seal(self):
        """no more files will be submitted for this frame"""
        self.file_done(None)

    def file_done(self, path, err=None):
        with self.lock:
            self.left -= 1
            if path is not None:
                self.paths.append(path)
            if err is not None and self.error is None:
                self.error = err
            last = self.left == 0
        if last and self.on_done is not None:
            try:
                self.on_done(self)
            finally:
                # the frame and its group point at each other: without this the frame's device tensors (its subtraction
                # images, 2.7 GB per frame) wait for the cyclic collector -- a list of a hundred frames ran the GPU out of memory
                self.token = self.on_done = None


class FzLane:
    """device buffers of one lane (a library context + its stream): one scratch for the tile streams at their stride
    (reused image after image: stream order), [nslots] output slots that hold table + heap until a writer has copied them"""

    def __init__(self, ctx, ny, nx, nslots=12, heap_frac=0.6):
        self.ctx, self.ny, self.nx = ctx, ny, nx
        dev = ctx.device
        self.d_scratch = torch.empty(ny * lib.bbx_fpack_tile_stride(nx, 4), dtype=torch.uint8, device=dev)
        self.d_tiles = torch.empty(ny * 24, dtype=torch.uint8, device=dev)
        self.d_off = torch.empty(ny, dtype=torch.int64, device=dev)
        self.free = queue.Queue()
        self.slots = []
        self.slot_wait = 0.0
        cap = ny * 32 + int(heap_frac * ny * nx * 4) + 4096
        for _ in range(nslots):
            s = _Slot()
            s.cap = cap
            s.d_body = torch.empty(cap, dtype=torch.uint8, device=dev)
            s.d_info = torch.empty(4 + MAX_LIST, dtype=torch.int64, device=dev)
            s.h_info = torch.empty(4 + MAX_LIST, dtype=torch.int64, pin_memory=True)
            s.ev = torch.cuda.Event()
            self.slots.append(s)
            self.free.put(s)

    def enqueue(self, img, quant, seed, scale=1.0):
        """queue the compression of [img] (contiguous 2-D float32 / uint8 / int16 / int32 device tensor of this lane's
        shape) on the current stream -> slot (blocks only when all slots are still waiting for their writer)"""
        if tuple(img.shape) != (self.ny, self.nx) or not img.is_contiguous():
            raise ValueError('image of shape {} expected'.format((self.ny, self.nx)))
        bitpix = {torch.float32: -32, torch.uint8: 8, torch.int16: 16, torch.int32: 32}[img.dtype]
        t0 = time.perf_counter()
        s = self.free.get()
        self.slot_wait += time.perf_counter() - t0                # (lane thread: time it stood waiting for a free slot)
        rnd = fpack._rnd(img.device) if bitpix == -32 else None
        st = torch.cuda.current_stream(img.device)
        check(lib.bbx_fpack_body_scaled(self.ctx.h, self.ny, self.nx, C.c_void_p(img.data_ptr()), bitpix, float(quant), int(seed),
                                        C.c_void_p(rnd.data_ptr()) if rnd is not None else None, C.c_void_p(self.d_scratch.data_ptr()),
                                        C.c_void_p(self.d_tiles.data_ptr()), C.c_void_p(self.d_off.data_ptr()),
                                        C.c_void_p(s.d_body.data_ptr()), s.cap, C.c_void_p(s.d_info.data_ptr()), MAX_LIST, float(scale),
                                        C.c_void_p(st.cuda_stream)), 'bbx_fpack_body', self.ctx.h)
        # the summary by a kernel copy (the copy engines are busy with other images' 100 MB bodies)
        check(lib.bbx_copy_kernel(C.c_void_p(s.h_info.data_ptr()), C.c_void_p(s.d_info.data_ptr()), s.h_info.numel() * 8,
                                  C.c_void_p(st.cuda_stream)), 'bbx_copy_kernel')
        s.ev.record(st)
        return s, bitpix


class OutputStage:
    """writer pool + per-lane device buffers.  submit() is called by a lane thread (inside its torch.cuda.stream) right
    after the image has been queued; headers may come later (FrameGroup.header_ready)."""

    def __init__(self, device, ny, nx, nwriters=8, nslots=12, heap_frac=0.6, dither_seed=1):
        # nslots: at least the images of two frames per lane, so that a lane can always finish queueing its frame (a writer
        # that has copied a slot out may then wait for that frame's header, which only comes once the lane is through)
        self.device, self.ny, self.nx = device, ny, nx
        self.nslots, self.heap_frac, self.seed = nslots, heap_frac, dither_seed
        self.lanes = {}
        self.q = queue.Queue()
        self.threads = [threading.Thread(target=self._writer, args=(k,), daemon=True, name='bbx-writer') for k in range(nwriters)]
        self.bytes_written, self.files_written = 0, 0
        self.phase = {}                       # writer phases: name -> [wall s, cpu s, calls]
        self.stat_lock = threading.Lock()
        for t in self.threads:
            t.start()

    def lane(self, ctx):
        """the device buffers of the lane behind [ctx] (made at its first image, under its stream)"""
        key = id(ctx)
        if key not in self.lanes:
            self.lanes[key] = FzLane(ctx, self.ny, self.nx, self.nslots, self.heap_frac)
        return self.lanes[key]

    def new_group(self, token=None, on_done=None):
        return FrameGroup(token, on_done)

    def submit(self, ctx, group, img, path, quant=None, bzero=None, scale=1.0):
        """queue [img] -> [path].fz.  The caller keeps [img] unchanged until the group reports the file done.
        scale (float images): the file holds float32(scale) x img, multiplied as the kernel loads the pixels"""
        lane = self.lane(ctx)
        out = path if path.endswith('.fz') else path + '.fz'
        q = default_quant(path) if quant is None else quant
        if img.dtype == torch.uint16:                       # FITS stores uint16 as int16 with BZERO = 32768
            img = (img.to(torch.int32) - 32768).to(torch.int16)
            bzero = 32768
        scale = float(np.float32(scale))
        if scale != 1.0 and img.dtype != torch.float32:
            raise ValueError('scale: float32 images only')
        slot, bitpix = lane.enqueue(img, q, self.seed, scale)
        j = _Job()
        j.scale = scale
        j.lane, j.slot, j.img, j.path, j.quant, j.seed, j.bitpix, j.shape, j.group, j.bzero = lane, slot, img, out, q, self.seed, bitpix, \
            (self.ny, self.nx), group, bzero
        with group.lock:
            group.left += 1
        self.q.put(j)
        return out

    def close(self, timeout=60.0):
        """stop the writers.  Frames that never got their headers must have been cancelled by their owner
        (FrameGroup.cancel; FramePipeline does that when a run fails): a writer still waiting after [timeout] seconds
        in total is reported instead of being waited for one by one"""
        import time
        for _ in self.threads:
            self.q.put(None)
        t_end = time.monotonic() + timeout
        for t in self.threads:
            t.join(max(0.0, t_end - time.monotonic()))
        stuck = [t.name for t in self.threads if t.is_alive()]
        self.lanes.clear()
        if stuck:
            raise RuntimeError('output stage: {} writer thread(s) still waiting (a frame without headers that nobody cancelled?)'
                               .format(len(stuck)))

    # ---- writer thread ----------------------------------------------------------------------------
    def _writer(self, k):
        torch.cuda.set_device(self.device)
        copy_stream = torch.cuda.Stream(device=self.device)
        pinned = [None]

        def buf(n):
            if pinned[0] is None or pinned[0].numel() < n:
                pinned[0] = torch.empty(int(n * 1.2) + 65536, dtype=torch.uint8, pin_memory=True)
            return pinned[0][:n]
        while True:
            j = self.q.get()
            if j is None:
                return
            try:
                self._write(j, copy_stream, buf)
                j.group.file_done(j.path)
            except BaseException as e:                                     # reported with the frame
                j.group.file_done(j.path, e)

    def _tick(self, acc, name, t):
        """per-phase accounting of the writers: (wall, cpu) seconds since [t] -> acc[name]; -> new t"""
        now = (time.perf_counter(), time.thread_time())
        a = acc.setdefault(name, [0.0, 0.0])
        a[0] += now[0] - t[0]; a[1] += now[1] - t[1]
        return now

    def _write(self, j, copy_stream, buf):
        acc = {}
        try:
            self._write_timed(j, copy_stream, buf, acc)
        finally:
            with self.stat_lock:
                for k, (w, c) in acc.items():
                    a = self.phase.setdefault(k, [0.0, 0.0, 0])
                    a[0] += w; a[1] += c; a[2] += 1

    def _write_timed(self, j, copy_stream, buf, acc):
        ny, nx = j.shape
        quant = j.bitpix == -32
        rowlen = 32 if quant else 8
        s = j.slot
        t = (time.perf_counter(), time.thread_time())
        wait_event(s.ev)
        t = self._tick(acc, 'wait_image', t)
        info = s.h_info.numpy()
        total, nlist, overflow, maxlen = int(info[0]), int(info[1]), int(info[2]), int(info[3])
        if overflow:
            # heap larger than the slot (an image that hardly compresses) or too many refused rows: the serial path, with
            # its own buffers -- one call at a time in the process (fpack._SERIAL) -- and on THIS thread's stream, behind the
            # event that says the image is complete
            j.lane.free.put(s)
            header = self._header(j)
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(s.ev)
                fpack.fpack_image(_OnStream(j.lane.ctx, copy_stream), j.path, j.img if j.scale == 1.0 else j.img * j.scale, header, j.quant,
                                  j.seed)
            return
        nbody = ny * rowlen + total
        listed = np.sort(info[4:4 + nlist].copy()) if nlist else None
        # one pinned buffer: [table + heap | the rows the quantiser refused, as float32], one wait for both copies
        rbytes = nlist * nx * 4 if (nlist and quant) else 0
        roff = (nbody + 63) // 64 * 64
        full = buf(roff + rbytes)
        hb = full[:nbody]
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(s.ev)
            hb.copy_(s.d_body[:nbody], non_blocking=True)
            rows = None
            if rbytes:
                rows_pin = full[roff:roff + rbytes].view(torch.float32).view(nlist, nx)
                sel = j.img.index_select(0, torch.from_numpy(listed).to(j.img.device))
                rows_pin.copy_(sel if j.scale == 1.0 else sel * j.scale, non_blocking=True)
            cev = torch.cuda.Event()
            cev.record(copy_stream)
        wait_event(cev)
        t = self._tick(acc, 'copy_out', t)
        j.lane.free.put(s)                                                 # the device slot can take the next image
        body = hb.numpy()
        if rbytes:
            rows = rows_pin.numpy()
        parts, maxgz = [], 0
        if nlist:
            table = body[:ny * rowlen].view([('len', '>i4'), ('off', '>i4'), ('glen', '>i4'), ('goff', '>i4'), ('zscale', '>f8'), ('zzero', '>f8')])
            pos = total
            be = rows.astype('>f4')
            for k, r in enumerate(listed):
                g = fpack.gzip_row(be[k].tobytes())
                table['glen'][r], table['goff'][r] = len(g), pos
                pos += len(g)
                maxgz = max(maxgz, len(g))
                parts.append(g)
        pcount = total + sum(len(g) for g in parts)
        t = self._tick(acc, 'gzip_rows', t)
        header = self._header(j)
        t = self._tick(acc, 'wait_header', t)
        head = fpack.fz_header_bytes(j.shape, j.bitpix, pcount, maxlen, maxgz, header, j.seed, j.bzero)
        t = self._tick(acc, 'format_header', t)
        nb = nbody + (pcount - total)
        with open(j.path, 'wb') as f:
            f.write(head)
            f.write(memoryview(body))
            for g in parts:
                f.write(g)
            f.write(b'\0' * ((-nb) % fitsio.BLOCK))
        t = self._tick(acc, 'write_file', t)
        with self.stat_lock:
            self.bytes_written += len(head) + nb
            self.files_written += 1

    @staticmethod
    def _header(j):
        g = j.group
        g.header_ready.wait()
        if g.cancelled:
            raise GroupCancelled(j.path)
        h = g.headers.get(j.path)
        if h is None:
            h = g.headers.get(None)
        return h
