"""Minimal FITS image reader/writer (numpy only).

The reference reads frames with astropy (`read_hdulist`, faithful copy at
blackbox_slurm_google.py:1144-1253: last HDU, `.astype(dtype)`) and writes them with
`fits.writeto` (blackbox.py:1987-1990).  astropy is not part of the GPU box's Python,
so the hot path owns this small uncompressed-image implementation: BITPIX 8 / 16 / 32 /
-32 / -64 primary or IMAGE-extension HDUs, BZERO/BSCALE for the unsigned-16 raw frames.
Tile-compressed (.fz) files are out of scope.

Headers are plain dicts {KEY: (value, comment)} -- the same shape the stage functions
fill -- or {KEY: value}.
"""
import numpy as np

BLOCK = 2880
_BITPIX = {8: '>u1', 16: '>i2', 32: '>i4', 64: '>i8', -32: '>f4', -64: '>f8'}


def _parse_value(s):
    s = s.strip()
    if not s:
        return None
    if s[0] == "'":
        end = 1
        while True:
            end = s.find("'", end)
            if end < 0:
                return s[1:].rstrip()
            if end + 1 < len(s) and s[end + 1] == "'":
                end += 2
                continue
            break
        return s[1:end].replace("''", "'").rstrip()
    v = s.split('/')[0].strip()
    if v == 'T':
        return True
    if v == 'F':
        return False
    try:
        return int(v)
    except ValueError:
        pass
    try:
        return float(v.replace('D', 'E'))
    except ValueError:
        return v


def _read_header(f):
    cards = {}
    order = []
    while True:
        block = f.read(BLOCK)
        if len(block) < BLOCK:
            raise EOFError('truncated FITS header')
        done = False
        for i in range(0, BLOCK, 80):
            card = block[i:i + 80].decode('ascii', 'replace')
            key = card[:8].strip()
            if key == 'END':
                done = True
                break
            if not key or key in ('COMMENT', 'HISTORY') or card[8:10] != '= ':
                continue
            body = card[10:]
            val = _parse_value(body)
            comment = ''
            if isinstance(val, str) and body.lstrip().startswith("'"):
                rest = body[body.find("'", body.find("'") + 1) + 1:] if body.count("'") >= 2 else ''
                comment = rest.split('/', 1)[1].strip() if '/' in rest else ''
            elif '/' in body:
                comment = body.split('/', 1)[1].strip()
            cards[key] = (val, comment)
            order.append(key)
        if done:
            break
    return cards


def _hv(h, k, default=None):
    v = h.get(k, default)
    return v[0] if isinstance(v, tuple) else v


def read_hdus(path, headers_only=False):
    """-> list of (header dict, data ndarray or None).  Binary tables come back as their raw
    row bytes (uint8 [NAXIS2, NAXIS1]) with the heap under header key '__heap__'.
    headers_only: the data units are skipped (data None)."""
    out = []
    with open(path, 'rb') as f:
        while True:
            pos = f.tell()
            probe = f.read(1)
            if not probe:
                break
            f.seek(pos)
            h = _read_header(f)
            naxis = _hv(h, 'NAXIS', 0)
            bitpix = _hv(h, 'BITPIX', 8)
            data = None
            if naxis > 0:
                shape = tuple(int(_hv(h, 'NAXIS%d' % k)) for k in range(naxis, 0, -1))
                npix = int(np.prod(shape))
                if headers_only:
                    nbytes = npix * abs(bitpix) // 8 + int(_hv(h, 'PCOUNT', 0) or 0)
                    f.seek(nbytes + (-nbytes) % BLOCK, 1)
                    out.append((h, None))
                    continue
                raw = np.fromfile(f, dtype=_BITPIX[bitpix], count=npix)
                if raw.size != npix:
                    raise EOFError('truncated FITS data in {}'.format(path))
                data = raw.reshape(shape)
                nbytes = npix * abs(bitpix) // 8
                pcount = int(_hv(h, 'PCOUNT', 0) or 0)
                if pcount and str(_hv(h, 'XTENSION', '')).strip() == 'BINTABLE':
                    h['__heap__'] = np.fromfile(f, dtype=np.uint8, count=pcount)
                    nbytes += pcount
                f.seek((-nbytes) % BLOCK, 1)                          # data units are padded to 2880 bytes
            out.append((h, data))
    return out


def read_image(path, dtype=None, get_header=False):
    """last HDU with data, like zogy.read_hdulist; unsigned-16 raws (BITPIX 16, BZERO 32768)
    come back as uint16, everything else scaled by BSCALE/BZERO when present."""
    hdus = read_hdus(path)
    h, data = None, None
    for hh, dd in hdus:
        if dd is not None:
            h, data = hh, dd
    if data is None:
        raise ValueError('no image data in {}'.format(path))
    bzero = _hv(h, 'BZERO', 0) or 0
    bscale = _hv(h, 'BSCALE', 1) or 1
    if data.dtype == np.dtype('>i2') and bzero == 32768 and bscale == 1:
        data = (data.astype(np.int32) + 32768).astype(np.uint16)
    elif bzero != 0 or bscale != 1:
        data = data.astype(np.float64) * bscale + bzero
    else:
        data = data.astype(data.dtype.newbyteorder('='))
    if dtype is not None:
        data = data.astype(dtype, copy=False)
    return (data, h) if get_header else data


def read_image_file_order(path):
    """last HDU with data as it lies in the file: (array in the file's big-endian dtype -- a view, nothing converted --,
    header) when no BSCALE / BZERO applies, else None (the caller takes read_image).  For images that go to the device,
    which swaps the bytes there (bbx_be32)."""
    h, data = None, None
    for hh, dd in read_hdus(path):
        if dd is not None:
            h, data = hh, dd
    if data is None:
        raise ValueError('no image data in {}'.format(path))
    if (_hv(h, 'BZERO', 0) or 0) != 0 or (_hv(h, 'BSCALE', 1) or 1) != 1 or 'XTENSION' in h and str(_hv(h, 'XTENSION', '')).strip() == 'BINTABLE':
        return None
    return data, h


def _card(key, value, comment=''):
    if isinstance(value, tuple):
        value, comment = value[0], (value[1] if len(value) > 1 else comment)
    key = str(key).upper()[:8]
    if isinstance(value, (bool, np.bool_)):
        v = '{:>20}'.format('T' if value else 'F')
    elif isinstance(value, (int, np.integer)):
        v = '{:>20d}'.format(int(value))
    elif isinstance(value, (float, np.floating)):
        if not np.isfinite(value):
            v = "'{:<8}'".format(str(value))
        else:
            v = '{:>20}'.format(repr(float(value)).upper().replace('E+', 'E'))
            if len(v) > 20:
                v = '{:>20.13G}'.format(float(value))
    else:
        s = str(value).replace("'", "''")[:68]
        v = "'{:<8}'".format(s)
    card = '{:<8}= {}'.format(key, v)
    if comment:
        card += ' / ' + str(comment)
    return card[:80].ljust(80)


def write_image(path, data, header=None):
    """uncompressed primary-HDU image: uint8 -> BITPIX 8, uint16 -> BITPIX 16 + BZERO,
    float32 -> BITPIX -32 (what blackbox.py:1987-1990 writes for _red and _mask)"""
    data = np.asarray(data)
    extra = []
    if data.dtype == np.uint8:
        bitpix, out = 8, data
    elif data.dtype == np.uint16:
        bitpix, out = 16, (data.astype(np.int32) - 32768).astype('>i2')
        extra = [('BSCALE', 1, ''), ('BZERO', 32768, '')]
    elif data.dtype == np.int16:
        bitpix, out = 16, data.astype('>i2')
    elif data.dtype == np.int32:
        bitpix, out = 32, data.astype('>i4')
    elif data.dtype == np.float64:
        bitpix, out = -64, data.astype('>f8')
    else:
        bitpix, out = -32, data.astype('>f4')
    cards = [_card('SIMPLE', True, 'conforms to FITS standard'), _card('BITPIX', bitpix, 'array data type'),
             _card('NAXIS', data.ndim, 'number of array dimensions')]
    for k in range(data.ndim):
        cards.append(_card('NAXIS%d' % (k + 1), data.shape[data.ndim - 1 - k]))
    for k, v, c in extra:
        cards.append(_card(k, v, c))
    skip = {'SIMPLE', 'BITPIX', 'NAXIS', 'EXTEND', 'BZERO', 'BSCALE', 'END'}
    for k, v in (header or {}).items():
        ku = str(k).upper()
        if ku in skip or ku.startswith('NAXIS') or len(ku) > 8:
            continue
        cards.append(_card(ku, v))
    cards.append('END'.ljust(80))
    hdr = ''.join(cards).encode('ascii', 'replace')
    hdr += b' ' * ((-len(hdr)) % BLOCK)
    with open(path, 'wb') as f:
        f.write(hdr)
        buf = out.tobytes()
        f.write(buf)
        f.write(b'\0' * ((-len(buf)) % BLOCK))


def _header_bytes(cards):
    hdr = ''.join(cards + ['END'.ljust(80)]).encode('ascii', 'replace')
    return hdr + b' ' * ((-len(hdr)) % BLOCK)


def _user_cards(header, skip=()):
    skip = set(skip) | {'SIMPLE', 'BITPIX', 'NAXIS', 'EXTEND', 'BZERO', 'BSCALE', 'END', 'XTENSION', 'PCOUNT', 'GCOUNT',
                        'TFIELDS'}
    cards = []
    for k, v in (header or {}).items():
        ku = str(k).upper()
        if ku in skip or ku.startswith('NAXIS') or ku.startswith('TFORM') or ku.startswith('TTYPE') or len(ku) > 8 \
                or ku.startswith('__'):
            continue
        cards.append(_card(ku, v))
    return cards


def write_header(path, header):
    """header-only FITS file: what update_imcathead(..., create_hdrfile=True) leaves next to the
    product (`_red_hdr.fits`, `_trans_hdr.fits`; blackbox.py:2011, zogy.update_imcathead)"""
    cards = [_card('SIMPLE', True, 'conforms to FITS standard'), _card('BITPIX', 8, 'array data type'),
             _card('NAXIS', 0, 'number of array dimensions')] + _user_cards(header)
    with open(path, 'wb') as f:
        f.write(_header_bytes(cards))


_TFORM = {'f4': 'E', 'f8': 'D', 'i2': 'I', 'i4': 'J', 'i8': 'K', 'u1': 'B', 'b1': 'L'}


def write_table(path, columns, header=None, units=None):
    """binary-table FITS file (empty primary HDU + one BINTABLE): columns = {name: 1-D or 2-D
    array}; what zogy.format_cat writes for `_cat.fits` / `_trans.fits`.  Zero rows is valid
    (the dummy catalogues of qc.py:451-503)."""
    names = list(columns)
    arrs = [np.asarray(columns[n]) for n in names]
    nrow = arrs[0].shape[0] if arrs else 0
    fields, forms = [], []
    for n, a in zip(names, arrs):
        if a.shape[0] != nrow:
            raise ValueError('column {} has {} rows, expected {}'.format(n, a.shape[0], nrow))
        code = a.dtype.str[1:]
        if code not in _TFORM:
            raise TypeError('column {}: dtype {} not supported'.format(n, a.dtype))
        rep = int(np.prod(a.shape[1:])) if a.ndim > 1 else 1
        fields.append((n, '>' + code if code != 'b1' else 'u1', (rep,) if rep > 1 else ()))
        forms.append(('%d' % rep if rep > 1 else '') + _TFORM[code])
    rec = np.zeros(nrow, dtype=[(n, t, s) for n, t, s in fields]) if fields else np.zeros(0, 'u1')
    for (n, _, _), a in zip(fields, arrs):
        if a.dtype == np.bool_:
            rec[n] = np.where(a.reshape(rec[n].shape), ord('T'), ord('F'))
        else:
            rec[n] = a.reshape(rec[n].shape)
    rowbytes = rec.dtype.itemsize if fields else 0
    cards = [_card('XTENSION', 'BINTABLE', 'binary table extension'), _card('BITPIX', 8), _card('NAXIS', 2),
             _card('NAXIS1', rowbytes, 'width of table in bytes'), _card('NAXIS2', nrow, 'number of rows'),
             _card('PCOUNT', 0), _card('GCOUNT', 1), _card('TFIELDS', len(names), 'number of columns')]
    for i, (n, form) in enumerate(zip(names, forms)):
        cards.append(_card('TTYPE%d' % (i + 1), n))
        cards.append(_card('TFORM%d' % (i + 1), form))
        if units and units.get(n):
            cards.append(_card('TUNIT%d' % (i + 1), units[n]))
    cards += _user_cards(header)
    prim = [_card('SIMPLE', True, 'conforms to FITS standard'), _card('BITPIX', 8), _card('NAXIS', 0),
            _card('EXTEND', True)]
    with open(path, 'wb') as f:
        f.write(_header_bytes(prim))
        f.write(_header_bytes(cards))
        buf = rec.tobytes()
        f.write(buf)
        f.write(b'\0' * ((-len(buf)) % BLOCK))


def read_table(path, ext=1):
    """-> (columns dict name -> native-endian array, header) of a BINTABLE HDU written with
    fixed-width columns (L B I J K E D with repeat counts)"""
    h, data = read_hdus(path)[ext]
    nf = int(_hv(h, 'TFIELDS', 0))
    nrow = int(_hv(h, 'NAXIS2', 0))
    code = {v: k for k, v in _TFORM.items()}
    fields = []
    for i in range(1, nf + 1):
        form = str(_hv(h, 'TFORM%d' % i)).strip()
        rep = int(form[:-1]) if form[:-1] else 1
        t = code[form[-1]]
        fields.append((str(_hv(h, 'TTYPE%d' % i)).strip(), ('>' + t) if t != 'b1' else 'u1', (rep,) if rep > 1 else ()))
    rec = np.frombuffer(data.tobytes() if data is not None else b'', dtype=fields, count=nrow) if fields else None
    cols = {}
    for n, t, s in fields:
        a = rec[n]
        cols[n] = (a == ord('T')) if t == 'u1' and str(_hv(h, 'TFORM%d' % (1 + [f[0] for f in fields].index(n)))).strip().endswith('L') \
            else a.astype(a.dtype.newbyteorder('='))
    return cols, h


def read_psfex(path):
    """PSFEx `.psf` file (binary table with one PSF_MASK cell; zogy.extract_psf_datapars,
    signature seen at buildref.py:3357-3366) -> dict(basis float32 [ncoef, S, S], polzero,
    polscal, poldeg, psf_samp, psf_fwhm)"""
    for h, data in read_hdus(path)[1:]:
        if str(_hv(h, 'XTENSION', '')).strip() != 'BINTABLE':
            continue
        form = str(_hv(h, 'TFORM1')).strip()
        n = int(form[:-1])
        dims = str(_hv(h, 'TDIM1', '')).strip('() ').split(',')
        shape = tuple(int(d) for d in reversed(dims)) if dims != [''] else (n,)
        basis = np.frombuffer(data.tobytes(), dtype='>f4', count=n).astype(np.float32).reshape(shape)
        ngroup = int(_hv(h, 'POLNGRP', 1) or 0)
        poldeg = int(_hv(h, 'POLDEG1', 0) or 0) if ngroup else 0
        return dict(basis=basis, poldeg=poldeg,
                    polzero=(float(_hv(h, 'POLZERO1', 0.0) or 0.0), float(_hv(h, 'POLZERO2', 0.0) or 0.0)),
                    polscal=(float(_hv(h, 'POLSCAL1', 1.0) or 1.0), float(_hv(h, 'POLSCAL2', 1.0) or 1.0)),
                    psf_samp=float(_hv(h, 'PSF_SAMP', 1.0) or 1.0), psf_fwhm=float(_hv(h, 'PSF_FWHM', 0.0) or 0.0))
    raise ValueError('no PSF_MASK table in {}'.format(path))
