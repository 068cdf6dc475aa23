"""fpack on the device (blackbox.py:812-857): tile-compressed FITS (`.fits.fz`) of the reduced
float image (`fpack -q 16`; -q 4 for Fpsf, -q 2 for Scorr / limmag) and of the uint8 mask
(lossless), RICE_1 with one tile per row and SUBTRACTIVE_DITHER_1, compressed by
`bbx_fpack_tiles` so that only the compressed bytes (~1/5 of the float frame, ~1/50 of the
mask) are copied to the host.  The binary-table container is assembled here."""
import ctypes as C

import numpy as np
import torch

from . import fitsio
from ._lib import lib, check

N_RANDOM = 10000
_TILE_DT = np.dtype([('nbytes', '<u4'), ('flag', '<u4'), ('zscale', '<f8'), ('zzero', '<f8')])


def fits_randoms():
    """CFITSIO's fits_init_randoms table (Park-Miller, seed 1): 10000 float32 in (0, 1)"""
    a, m, seed = 16807.0, 2147483647.0, 1.0
    out = np.empty(N_RANDOM, np.float32)
    for i in range(N_RANDOM):
        temp = a * seed
        seed = temp - m * int(temp / m)
        out[i] = np.float32(seed / m)
    if int(seed) != 1043618065:
        raise AssertionError('random table check value')
    return out


_RND = {}


def _rnd(device):
    t = _RND.get(str(device))
    if t is None:
        t = _RND[str(device)] = torch.from_numpy(fits_randoms()).to(device)
    return t


_PIN = {}


def _pinned(nbytes):
    """grow-only pinned staging buffer (hipHostMalloc costs ~0.1 ms per MB: not per call)"""
    t = _PIN.get('buf')
    if t is None or t.numel() < nbytes:
        t = _PIN['buf'] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
    return t[:nbytes]


def gzip_row(row_bytes):
    """a tile CFITSIO would not quantise, as it goes into the GZIP_COMPRESSED_DATA column: gzip of the big-endian floats.
    Level 1 like CFITSIO's compress2mem_from_mem (deflateInit2(.., 1, ..)): these are the constant edge rows, and the writer
    threads of the output stage compress 300 of them per frame"""
    import gzip
    return gzip.compress(row_bytes, 1, mtime=0)


def compress_tiles(ctx, img, qlevel=16, dither_seed=1, _view=False):
    """img: contiguous 2-D device tensor (float32 -> quantised; uint8 / int16 / int32 ->
    lossless).  -> (heap bytes numpy uint8 [total], tiles numpy structured [ny]: nbytes,
    flag, zscale, zzero; offsets numpy int64 [ny])"""
    if img.dim() != 2 or not img.is_contiguous():
        raise ValueError('contiguous 2-D image expected')
    if img.dtype == torch.uint16:                       # FITS stores uint16 as int16 with BZERO = 32768
        img = (img.to(torch.int32) - 32768).to(torch.int16)
    bitpix = {torch.float32: -32, torch.uint8: 8, torch.int16: 16, torch.int32: 32}[img.dtype]
    bytepix = 4 if bitpix == -32 else bitpix // 8
    ny, nx = img.shape
    stride = lib.bbx_fpack_tile_stride(nx, bytepix)
    dev = img.device
    scratch = torch.empty(ny * stride, dtype=torch.uint8, device=dev)
    tiles = torch.empty(ny * _TILE_DT.itemsize, dtype=torch.uint8, device=dev)
    rnd = _rnd(dev) if bitpix == -32 else None
    check(lib.bbx_fpack_tiles(ctx.h, ny, nx, C.c_void_p(img.data_ptr()), bitpix, float(qlevel), int(dither_seed),
                              C.c_void_p(rnd.data_ptr()) if rnd is not None else None, C.c_void_p(scratch.data_ptr()),
                              C.c_void_p(tiles.data_ptr()), ctx.stream()), 'bbx_fpack_tiles', ctx.h)
    t = tiles.cpu().numpy().view(_TILE_DT).copy()
    offsets = np.concatenate([[0], np.cumsum(t['nbytes'].astype(np.int64))])
    total = int(offsets[-1])
    d_off = torch.from_numpy(offsets[:-1].copy()).to(dev)
    heap = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
    check(lib.bbx_fpack_gather(ctx.h, ny, nx, bitpix, C.c_void_p(scratch.data_ptr()), C.c_void_p(tiles.data_ptr()),
                               C.c_void_p(d_off.data_ptr()), C.c_void_p(heap.data_ptr()), ctx.stream()),
          'bbx_fpack_gather', ctx.h)
    hp = _pinned(total)
    hp.copy_(heap[:total], non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    heap = hp.numpy() if _view else hp.numpy().copy()       # _view: valid until the next call (fpack_image writes at once)
    # rows that cannot be quantised (zero noise: constant rows such as the filled edge; or a
    # non-finite pixel) are stored losslessly, gzip of the big-endian floats, in the
    # GZIP_COMPRESSED_DATA column with ZSCALE = ZZERO = 0 -- as CFITSIO does
    gz_nbytes = np.zeros(ny, np.int64)
    gz_offsets = np.zeros(ny, np.int64)
    bad = np.nonzero(t['flag'])[0]
    if bad.size:
        import gzip
        rows = img[torch.from_numpy(bad).to(dev)].cpu().numpy().astype('>f4')
        parts, pos = [], total
        for k, r in enumerate(bad):
            g = gzip_row(rows[k].tobytes())
            gz_nbytes[r], gz_offsets[r] = len(g), pos
            pos += len(g)
            parts.append(np.frombuffer(g, np.uint8))
            t['zscale'][r] = t['zzero'][r] = 0.0
        heap = np.concatenate([heap] + parts)
    return dict(heap=heap, nbytes=t['nbytes'].astype(np.int64), offsets=offsets[:-1], zscale=t['zscale'], zzero=t['zzero'],
                flag=t['flag'], gz_nbytes=gz_nbytes, gz_offsets=gz_offsets)


def fz_header_bytes(shape, bitpix, pcount, maxlen, maxgz=0, header=None, dither_seed=1, bzero=None):
    """primary HDU + header of the COMPRESSED_IMAGE binary table (FITS 4.0 section 10), padded to blocks"""
    ny, nx = shape
    quant = bitpix == -32
    rowlen = 8 + (24 if quant else 0)
    cards = [fitsio._card('XTENSION', 'BINTABLE', 'binary table extension'), fitsio._card('BITPIX', 8, 'array data type'),
             fitsio._card('NAXIS', 2, 'number of array dimensions'), fitsio._card('NAXIS1', rowlen, 'width of table in bytes'),
             fitsio._card('NAXIS2', ny, 'number of rows in table'), fitsio._card('PCOUNT', int(pcount), 'size of the heap'),
             fitsio._card('GCOUNT', 1, 'number of groups'), fitsio._card('TFIELDS', 4 if quant else 1, 'number of fields in each row'),
             fitsio._card('TTYPE1', 'COMPRESSED_DATA', 'label for field 1'),
             fitsio._card('TFORM1', '1PB({})'.format(int(maxlen)), 'data format of field: variable length array')]
    if quant:
        cards += [fitsio._card('TTYPE2', 'GZIP_COMPRESSED_DATA', 'label for field 2'),
                  fitsio._card('TFORM2', '1PB({})'.format(int(maxgz)), 'data format of field: variable length array'),
                  fitsio._card('TTYPE3', 'ZSCALE', 'label for field 3'), fitsio._card('TFORM3', '1D', 'data format of field: 8-byte DOUBLE'),
                  fitsio._card('TTYPE4', 'ZZERO', 'label for field 4'), fitsio._card('TFORM4', '1D', 'data format of field: 8-byte DOUBLE')]
    cards += [fitsio._card('ZIMAGE', True, 'extension contains compressed image'),
              fitsio._card('ZTENSION', 'IMAGE', 'Image extension'), fitsio._card('ZBITPIX', bitpix, 'data type of original image'),
              fitsio._card('ZNAXIS', 2, 'dimension of original image'), fitsio._card('ZNAXIS1', nx, 'length of original image axis'),
              fitsio._card('ZNAXIS2', ny, 'length of original image axis'), fitsio._card('ZPCOUNT', 0, 'number of parameters'),
              fitsio._card('ZGCOUNT', 1, 'number of groups'), fitsio._card('ZTILE1', nx, 'size of tiles to be compressed'),
              fitsio._card('ZTILE2', 1, 'size of tiles to be compressed'), fitsio._card('ZCMPTYPE', 'RICE_1', 'compression algorithm'),
              fitsio._card('ZNAME1', 'BLOCKSIZE', 'compression block size'), fitsio._card('ZVAL1', 32, 'pixels per block'),
              fitsio._card('ZNAME2', 'BYTEPIX', 'bytes per pixel (1, 2, 4, or 8)'),
              fitsio._card('ZVAL2', 4 if quant else bitpix // 8, 'bytes per pixel (1, 2, 4, or 8)')]
    if quant:
        cards += [fitsio._card('ZQUANTIZ', 'SUBTRACTIVE_DITHER_1', 'Pixel Quantization Algorithm'),
                  fitsio._card('ZDITHER0', int(dither_seed), 'dithering offset when quantizing floats')]
    if bzero:
        cards += [fitsio._card('BSCALE', 1, ''), fitsio._card('BZERO', int(bzero), 'offset data range to that of unsigned short')]
    cards.append(fitsio._card('EXTNAME', 'COMPRESSED_IMAGE', 'name of this binary table extension'))
    skip = {'SIMPLE', 'BITPIX', 'NAXIS', 'EXTEND', 'BZERO', 'BSCALE', 'END', 'XTENSION', 'PCOUNT', 'GCOUNT', 'TFIELDS', 'EXTNAME'}
    for k, v in (header or {}).items():
        ku = str(k).upper()
        if ku in skip or ku.startswith('NAXIS') or ku.startswith('Z') and ku[1:4] in ('IMA', 'TEN', 'BIT', 'NAX', 'PCO', 'GCO', 'TIL', 'CMP', 'NAM', 'VAL', 'QUA', 'DIT') \
                or ku.startswith('TTYPE') or ku.startswith('TFORM') or len(ku) > 8:
            continue
        cards.append(fitsio._card(ku, v))
    cards.append('END'.ljust(80))
    ext = ''.join(cards).encode('ascii', 'replace')
    ext += b' ' * ((-len(ext)) % fitsio.BLOCK)
    prim = ''.join([fitsio._card('SIMPLE', True, 'conforms to FITS standard'), fitsio._card('BITPIX', 8, 'array data type'),
                    fitsio._card('NAXIS', 0, 'number of array dimensions'), fitsio._card('EXTEND', True),
                    'END'.ljust(80)]).encode('ascii')
    prim += b' ' * ((-len(prim)) % fitsio.BLOCK)
    return prim + ext


def assemble_fz(path, shape, bitpix, heap, nbytes, offsets, zscale=None, zzero=None, header=None, qlevel=16,
                dither_seed=1, gz_nbytes=None, gz_offsets=None, bzero=None):
    """write primary HDU + COMPRESSED_IMAGE binary table (FITS 4.0 section 10) around tile
    streams that are already compressed"""
    ny, nx = shape
    quant = bitpix == -32
    maxlen = int(np.max(nbytes)) if len(nbytes) else 0
    if gz_nbytes is None:
        gz_nbytes, gz_offsets = np.zeros(ny, np.int64), np.zeros(ny, np.int64)
    maxgz = int(np.max(gz_nbytes)) if len(gz_nbytes) else 0
    head = fz_header_bytes(shape, bitpix, len(heap), maxlen, maxgz, header, dither_seed, bzero)
    if quant:
        rows = np.zeros(ny, dtype=[('len', '>i4'), ('off', '>i4'), ('glen', '>i4'), ('goff', '>i4'), ('zscale', '>f8'),
                                   ('zzero', '>f8')])
        rows['zscale'], rows['zzero'] = zscale, zzero
        rows['glen'], rows['goff'] = gz_nbytes, gz_offsets
    else:
        rows = np.zeros(ny, dtype=[('len', '>i4'), ('off', '>i4')])
    rows['len'], rows['off'] = nbytes, offsets
    body = rows.tobytes() + bytes(heap)
    with open(path, 'wb') as f:
        f.write(head)
        f.write(body)
        f.write(b'\0' * ((-len(body)) % fitsio.BLOCK))
    return path


_BODY = {}
# fpack_image works in process-wide device / pinned buffers (_BODY, _PIN): one call at a time.  Callers on several threads
# (the output stage's writers in their overflow fallback, the orchestrating thread writing an unstaged product) queue here.
_SERIAL = __import__('threading').RLock()


def _body_buffers(dev, ny, nx, need=None):
    """device buffers of the one-enqueue path (bbx_fpack_body), kept per device and shape"""
    key = (str(dev), ny, nx)
    b = _BODY.get(key)
    cap = ny * 32 + (int(0.6 * ny * nx * 4) if need is None else int(need)) + 4096
    if b is None or b['cap'] < cap:
        b = _BODY[key] = dict(cap=cap, scratch=torch.empty(ny * lib.bbx_fpack_tile_stride(nx, 4), dtype=torch.uint8, device=dev),
                              tiles=torch.empty(ny * 24, dtype=torch.uint8, device=dev), off=torch.empty(ny, dtype=torch.int64, device=dev),
                              body=torch.empty(cap, dtype=torch.uint8, device=dev),
                              info=torch.empty(4 + 4096, dtype=torch.int64, device=dev))
    return b


def fpack_image(ctx, path, img, header=None, quant=None, dither_seed=1):
    """the reference's fpack(filename) for a device image: float32 -> quantised with level
    [quant] (default by product name: 2 for Scorr / limmag, 4 for Fpsf, else 16); integer ->
    lossless.  -> path of the .fz file.  One enqueue on the device (tile streams, offsets, descriptor table:
    bbx_fpack_body), one copy of exactly the bytes of the file body, the rows the quantiser refused gzip-compressed
    on the host as CFITSIO stores them."""
    with _SERIAL:
        return _fpack_image(ctx, path, img, header, quant, dither_seed)


def _fpack_image(ctx, path, img, header, quant, dither_seed):
    if quant is None:
        quant = 2 if ('Scorr' in path or 'limmag' in path) else (4 if 'Fpsf' in path else 16)
    out = path if path.endswith('.fz') else path + '.fz'
    if img.dim() != 2 or not img.is_contiguous():
        raise ValueError('contiguous 2-D image expected')
    bzero = None
    if img.dtype == torch.uint16:                       # FITS stores uint16 as int16 with BZERO = 32768
        img = (img.to(torch.int32) - 32768).to(torch.int16)
        bzero = 32768
    bitpix = {torch.float32: -32, torch.uint8: 8, torch.int16: 16, torch.int32: 32}[img.dtype]
    ny, nx = img.shape
    dev = img.device
    rowlen = 32 if bitpix == -32 else 8
    need = None
    # (the kernels run on ctx.stream(); the copies below on torch's current stream: the two must be the same stream)
    for _ in range(2):
        b = _body_buffers(dev, ny, nx, need)
        rnd = _rnd(dev) if bitpix == -32 else None
        check(lib.bbx_fpack_body(ctx.h, ny, nx, C.c_void_p(img.data_ptr()), bitpix, float(quant), int(dither_seed),
                                 C.c_void_p(rnd.data_ptr()) if rnd is not None else None, C.c_void_p(b['scratch'].data_ptr()),
                                 C.c_void_p(b['tiles'].data_ptr()), C.c_void_p(b['off'].data_ptr()), C.c_void_p(b['body'].data_ptr()),
                                 b['cap'], C.c_void_p(b['info'].data_ptr()), 4096, ctx.stream()), 'bbx_fpack_body', ctx.h)
        info = b['info'].cpu().numpy()
        total, nlist, overflow, maxlen = int(info[0]), int(info[1]), int(info[2]), int(info[3])
        if not overflow:
            break
        if nlist > 4096:
            raise ValueError('{} rows of the image cannot be quantised'.format(nlist))
        need = total                                    # an image that hardly compresses: once more with room for it
    nbody = ny * rowlen + total
    hp = _pinned(nbody)
    hp.copy_(b['body'][:nbody], non_blocking=True)
    parts, maxgz = [], 0
    listed = np.sort(info[4:4 + nlist]) if nlist else None
    rows = img[torch.from_numpy(listed).to(dev)].cpu().numpy().astype('>f4') if nlist else None
    torch.cuda.current_stream(dev).synchronize()
    body = hp.numpy()
    if nlist:
        import gzip
        table = body[:ny * rowlen].view([('len', '>i4'), ('off', '>i4'), ('glen', '>i4'), ('goff', '>i4'), ('zscale', '>f8'), ('zzero', '>f8')])
        pos = total
        for k, r in enumerate(listed):
            g = gzip_row(rows[k].tobytes())
            table['glen'][r], table['goff'][r] = len(g), pos
            pos += len(g)
            maxgz = max(maxgz, len(g))
            parts.append(g)
    pcount = total + sum(len(g) for g in parts)
    head = fz_header_bytes((ny, nx), bitpix, pcount, maxlen, maxgz, header, dither_seed, bzero)
    nb = nbody + pcount - total
    with open(out, 'wb') as f:
        f.write(head)
        f.write(memoryview(body))
        for g in parts:
            f.write(g)
        f.write(b'\0' * ((-nb) % fitsio.BLOCK))
    return out


def fpack_image_serial(ctx, path, img, header=None, quant=None, dither_seed=1):
    """the same file through the step-by-step path (compress_tiles + assemble_fz): kept as the cross-check of the
    one-enqueue path (tests/test_fpack.py: byte-identical files)"""
    if quant is None:
        quant = 2 if ('Scorr' in path or 'limmag' in path) else (4 if 'Fpsf' in path else 16)
    out = path if path.endswith('.fz') else path + '.fz'
    c = compress_tiles(ctx, img, quant, dither_seed, _view=True)
    bitpix = {torch.float32: -32, torch.uint8: 8, torch.int16: 16, torch.int32: 32, torch.uint16: 16}[img.dtype]
    return assemble_fz(out, tuple(img.shape), bitpix, c['heap'], c['nbytes'], c['offsets'], c['zscale'], c['zzero'], header,
                       quant, dither_seed, c['gz_nbytes'], c['gz_offsets'], bzero=32768 if img.dtype == torch.uint16 else None)


# --------------------------------------------------------------------------------
# reading: funpack on the device
# --------------------------------------------------------------------------------
_TFORM_BYTES = {'L': 1, 'X': 1, 'B': 1, 'I': 2, 'J': 4, 'K': 8, 'A': 1, 'E': 4, 'D': 8, 'C': 8, 'M': 16}


def _table_layout(h):
    """column name -> (byte offset in the row, TFORM) of a binary table header"""
    import re
    hv = fitsio._hv
    out, off = {}, 0
    for k in range(1, int(hv(h, 'TFIELDS')) + 1):
        form = str(hv(h, 'TFORM%d' % k)).strip()
        name = str(hv(h, 'TTYPE%d' % k, 'COL%d' % k)).strip()
        m = re.match(r'(\d*)([PQ]?)([A-Z])', form)
        rep = int(m.group(1)) if m.group(1) else 1
        if m.group(2) == 'P':
            size = 8 * rep
        elif m.group(2) == 'Q':
            size = 16 * rep
        else:
            size = rep * _TFORM_BYTES[m.group(3)]
        out[name] = (off, form)
        off += size
    return out


def funpack_image(ctx, path, ext=None):
    """read a tile-compressed image (what the reference gets from read_hdulist on a .fits.fz
    file): RICE_1, tiles = whole rows, integer images (BITPIX 8/16/32; 16 with BZERO 32768 ->
    uint16) and float images quantised with SUBTRACTIVE_DITHER_1; rows in the
    GZIP_COMPRESSED_DATA column are inflated on the host.  -> (device tensor, header dict)"""
    import gzip
    hv = fitsio._hv
    hdus = fitsio.read_hdus(path)
    cand = [i for i, (h, d) in enumerate(hdus) if hv(h, 'ZIMAGE', False) is True] if ext is None else [ext]
    if not cand:
        raise ValueError('{}: no tile-compressed image extension'.format(path))
    h, table = hdus[cand[0]]
    heap = h.pop('__heap__', np.zeros(0, np.uint8))
    if str(hv(h, 'ZCMPTYPE')).strip() not in ('RICE_1', 'RICE_ONE'):
        raise ValueError('compression {} not supported (RICE_1 only)'.format(hv(h, 'ZCMPTYPE')))
    ny, nx, zbitpix = int(hv(h, 'ZNAXIS2')), int(hv(h, 'ZNAXIS1')), int(hv(h, 'ZBITPIX'))
    if int(hv(h, 'ZNAXIS')) != 2 or int(hv(h, 'ZTILE1', nx)) != nx or int(hv(h, 'ZTILE2', 1)) != 1:
        raise ValueError('only 2-D images tiled by rows are supported')
    zpar = {str(hv(h, 'ZNAME%d' % k)).strip(): hv(h, 'ZVAL%d' % k) for k in range(1, 5) if ('ZNAME%d' % k) in h}
    if int(zpar.get('BLOCKSIZE', 32)) != 32:
        raise ValueError('Rice block size {} not supported'.format(zpar.get('BLOCKSIZE')))
    bytepix = int(zpar.get('BYTEPIX', 4))
    lay = _table_layout(h)
    if 'COMPRESSED_DATA' not in lay or lay['COMPRESSED_DATA'][1][1] != 'P':
        raise ValueError('COMPRESSED_DATA column with 32-bit descriptors expected')
    tb = np.ascontiguousarray(table).reshape(ny, -1)
    o = lay['COMPRESSED_DATA'][0]
    desc = np.ascontiguousarray(tb[:, o:o + 8]).view('>i4').astype(np.int32)
    dev = ctx.device
    quant = zbitpix == -32
    bzero = hv(h, 'BZERO', 0)
    if quant:
        if str(hv(h, 'ZQUANTIZ', '')).strip() != 'SUBTRACTIVE_DITHER_1':
            raise ValueError('quantisation method {} not supported'.format(hv(h, 'ZQUANTIZ')))
        zs = np.ascontiguousarray(tb[:, lay['ZSCALE'][0]:lay['ZSCALE'][0] + 8]).view('>f8').astype(np.float64).reshape(-1)
        zz = np.ascontiguousarray(tb[:, lay['ZZERO'][0]:lay['ZZERO'][0] + 8]).view('>f8').astype(np.float64).reshape(-1)
        out = torch.empty((ny, nx), dtype=torch.float32, device=dev)
        kind, seed = 4, int(hv(h, 'ZDITHER0'))
        d_zs, d_zz, rnd = torch.from_numpy(zs).to(dev), torch.from_numpy(zz).to(dev), _rnd(dev)
    else:
        if zbitpix == 8:
            out, kind = torch.empty((ny, nx), dtype=torch.uint8, device=dev), 0
        elif zbitpix == 16 and bzero == 32768:
            out, kind = torch.empty((ny, nx), dtype=torch.uint16, device=dev), 1
        elif zbitpix == 16:
            out, kind = torch.empty((ny, nx), dtype=torch.int16, device=dev), 2
        elif zbitpix == 32:
            out, kind = torch.empty((ny, nx), dtype=torch.int32, device=dev), 3
        else:
            raise ValueError('ZBITPIX {} not supported'.format(zbitpix))
        if bytepix != abs(zbitpix) // 8:
            raise ValueError('BYTEPIX {} with ZBITPIX {} not supported'.format(bytepix, zbitpix))
        seed, d_zs, d_zz, rnd = 0, None, None, None
    if (desc < 0).any() or int((desc[:, 0].astype(np.int64) + desc[:, 1]).max(initial=0)) > heap.size:
        raise ValueError('tile descriptors point outside the heap')
    # compressed bytes + 16 readable pad bytes (no host-side copy of the heap for the padding)
    d_heap = torch.empty(heap.size + 16, dtype=torch.uint8, device=dev)
    d_heap[heap.size:].zero_()
    if heap.size:
        d_heap[:heap.size].copy_(torch.from_numpy(np.ascontiguousarray(heap)))
    d_desc = torch.from_numpy(desc.reshape(-1).copy()).to(dev)
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    check(lib.bbx_funpack_tiles(ctx.h, ny, nx, bytepix, vp(d_desc), vp(d_heap), kind, vp(out), vp(d_zs), vp(d_zz), seed,
                                vp(rnd), ctx.stream()), 'bbx_funpack_tiles', ctx.h)
    if quant and 'GZIP_COMPRESSED_DATA' in lay:
        og = lay['GZIP_COMPRESSED_DATA'][0]
        gd = np.ascontiguousarray(tb[:, og:og + 8]).view('>i4')
        for r in np.nonzero((desc[:, 0] == 0) & (gd[:, 0] > 0))[0]:
            raw = gzip.decompress(heap[gd[r, 1]:gd[r, 1] + gd[r, 0]].tobytes())
            out[r] = torch.from_numpy(np.frombuffer(raw, '>f4').astype(np.float32)).to(dev)
    ctx.sync()
    skip = ('ZIMAGE', 'ZCMPTYPE', 'ZBITPIX', 'ZNAXIS', 'ZTILE', 'ZNAME', 'ZVAL', 'ZQUANTIZ', 'ZDITHER0', 'ZTENSION', 'ZPCOUNT',
            'ZGCOUNT', 'TTYPE', 'TFORM', 'TFIELDS', 'XTENSION', 'PCOUNT', 'GCOUNT', 'NAXIS', 'BITPIX', 'EXTNAME')
    header = {k: v for k, v in h.items() if not any(k.startswith(p) for p in skip)}
    return out, header
