"""CPU: the host part of the input stage (blackbox_amd/instage.py, reference read_hdulist blackbox.py:1451): what parse_raw
finds in a raw frame file -- an fpacked `.fits.fz` (tile descriptors and heap position: the rows decode with the oracle's Rice
decoder from exactly the bytes the descriptors point at) and a plain `.fits` (data position); headers lose the structural
cards; malformed files are refused."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
import fpack as FP                                      # noqa: E402
from blackbox_amd import fitsio, instage                # noqa: E402
from blackbox_amd import fpack as P                     # noqa: E402


def test_parse_raw_fz_and_plain(tmp_path):
    rs = np.random.RandomState(5)
    ny, nx = 24, 333                                       # ragged last Rice block
    raw = (1500 + rs.normal(0, 9, (ny, nx))).astype(np.uint16)
    raw[3, 40:60] = 65535; raw[7] = 0
    i16 = (raw.astype(np.int32) - 32768).astype(np.int16)
    streams = [FP.rice_encode(i16[r], 2) for r in range(ny)]
    nbytes = np.array([len(s) for s in streams], np.int64)
    offsets = np.concatenate([[0], np.cumsum(nbytes)[:-1]])
    heap = np.frombuffer(b''.join(streams), np.uint8)
    hdr = {'EXPTIME': (60.0, 'exposure'), 'OBJECT': 'field 7', 'IMAGETYP': 'object', 'DATE-OBS': '2024-01-02T03:04:05'}
    pz = str(tmp_path / 'raw.fits.fz')
    P.assemble_fz(pz, (ny, nx), 16, heap, nbytes, offsets, header=hdr, bzero=32768)
    buf = np.fromfile(pz, np.uint8)
    r = instage.parse_raw(buf, buf.size)
    assert r.compressed and (r.ny, r.nx, r.bitpix, r.bytepix, r.bzero) == (ny, nx, 16, 2, 32768)
    assert r.heap_len == heap.size and r.heap_off % 4 == 0 and r.heap_off + r.heap_len <= buf.size
    assert np.array_equal(r.desc[:, 0], nbytes) and np.array_equal(r.desc[:, 1], offsets)
    for row in (0, 3, 7, ny - 1):
        ln, off = r.desc[row]
        got = FP.rice_decode(bytes(buf[r.heap_off + off:r.heap_off + off + ln]), nx, 2)
        assert np.array_equal((got.astype(np.int32) + 32768).astype(np.uint16), raw[row]), row
    assert fitsio._hv(r.header, 'OBJECT') == 'field 7' and fitsio._hv(r.header, 'EXPTIME') == 60.0
    assert not any(k.startswith(('Z', 'NAXIS', 'TFORM', 'TTYPE', 'BZERO', 'BITPIX', 'PCOUNT')) for k in r.header)
    # plain FITS
    pp = str(tmp_path / 'raw.fits')
    fitsio.write_image(pp, raw, hdr)
    buf = np.fromfile(pp, np.uint8)
    r = instage.parse_raw(buf, buf.size)
    assert not r.compressed and (r.ny, r.nx, r.bitpix, r.bzero) == (ny, nx, 16, 32768) and r.data_len == ny * nx * 2
    be = buf[r.data_off:r.data_off + r.data_len].view('>i2').reshape(ny, nx)
    assert np.array_equal((be.astype(np.int32) + 32768).astype(np.uint16), raw)
    assert fitsio._hv(r.header, 'IMAGETYP') == 'object' and 'BZERO' not in r.header
    # refused: descriptors beyond the heap, a truncated header, a file without an image
    bad = np.fromfile(pz, np.uint8).copy()
    tb = r_off = None
    rz = instage.parse_raw(bad, bad.size)
    table_off = rz.heap_off - ny * 8
    bad[table_off:table_off + 4] = np.frombuffer(np.array([10 ** 6], '>i4').tobytes(), np.uint8)      # first tile: a length beyond the heap
    with pytest.raises(ValueError):
        instage.parse_raw(bad, bad.size)
    with pytest.raises(EOFError):
        instage.parse_raw(buf[:1000], 1000)
    ph = str(tmp_path / 'hdr.fits')
    fitsio.write_header(ph, hdr)
    b3 = np.fromfile(ph, np.uint8)
    with pytest.raises(ValueError):
        instage.parse_raw(b3, b3.size)
    # truncated files -- still being written, a broken transfer -- parse as far as they go; their pixels would be decoded from
    # whatever the reader's buffer held before.  Refused like astropy refuses them: a heap cut short, a tile table cut
    # short, an uncompressed image cut short (nbytes = what was read; the buffer behind it may hold anything)
    bz = np.fromfile(pz, np.uint8)
    big = np.concatenate([bz, np.full(4096, 7, np.uint8)])
    rz = instage.parse_raw(big, bz.size)
    for cut in (rz.heap_off + rz.heap_len - 1, rz.heap_off + 5, rz.heap_off - 3):
        with pytest.raises(EOFError):
            instage.parse_raw(big, cut)
    bp = np.fromfile(pp, np.uint8)
    rp = instage.parse_raw(bp, bp.size)
    with pytest.raises(EOFError):
        instage.parse_raw(np.concatenate([bp, bp]), rp.data_off + rp.data_len - 2)
    assert instage.parse_raw(np.concatenate([bp, bp]), rp.data_off + rp.data_len).data_len == rp.data_len      # padding missing: the pixels are all there
