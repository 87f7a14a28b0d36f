"""Minimal FITS primary-HDU reader/writer (numpy only).

The reference does all I/O through ``astropy.io.fits`` (``fits.open(uint=True,
do_not_scale_image_data=False)`` / ``writeto``; e.g. core/ApCalibrate.py:260-328, 348-404).  astropy is
not available next to the GPU, and the hot path only ever touches the primary image HDU, so this
module implements exactly that subset:

* header: ordered 80-column cards with update/append/delete and HISTORY/COMMENT, value types
  bool / int / float / str, pass-through of unparsed cards;
* data: BITPIX 8, 16, 32, 64, -32, -64, big-endian on disk; the unsigned-integer convention
  (BITPIX=16, BSCALE=1, BZERO=32768 -> uint16; likewise uint32/uint64) as astropy's ``uint=True``;
  any other BSCALE/BZERO is applied like astropy does (float32 for <=16-bit integers, float64 above);
* extensions after the primary HDU are preserved verbatim on rewrite.
"""
import os

import numpy as np

BLOCK = 2880
_BITPIX_DTYPE = {8: '>u1', 16: '>i2', 32: '>i4', 64: '>i8', -32: '>f4', -64: '>f8'}
_COMMENTARY = ('HISTORY', 'COMMENT', '')


class Card:
    __slots__ = ('key', 'value', 'comment', 'raw')

    def __init__(self, key, value=None, comment='', raw=None):
        self.key = key
        self.value = value
        self.comment = comment
        self.raw = raw              # original 80-char image if the card was read and not modified

    def __repr__(self):
        return 'Card(%r, %r, %r)' % (self.key, self.value, self.comment)


def _parse_value(s):
    """Value/comment field (columns 11-80) of a FITS card -> (value, comment)."""
    s = s.rstrip()
    t = s.lstrip()
    if t.startswith("'"):
        # quoted string; '' is an escaped quote
        i = 1
        out = []
        while i < len(t):
            if t[i] == "'":
                if i + 1 < len(t) and t[i + 1] == "'":
                    out.append("'")
                    i += 2
                    continue
                break
            out.append(t[i])
            i += 1
        val = ''.join(out).rstrip()
        rest = t[i + 1:]
        comment = rest.split('/', 1)[1].strip() if '/' in rest else ''
        return val, comment
    if '/' in t:
        v, comment = t.split('/', 1)
        comment = comment.strip()
    else:
        v, comment = t, ''
    v = v.strip()
    if v == '':
        return None, comment
    if v == 'T':
        return True, comment
    if v == 'F':
        return False, comment
    try:
        return int(v), comment
    except ValueError:
        pass
    try:
        return float(v.replace('D', 'E').replace('d', 'e')), comment
    except ValueError:
        return v, comment


def _format_value(value):
    if isinstance(value, (bool, np.bool_)):
        return '%20s' % ('T' if value else 'F')
    if isinstance(value, (int, np.integer)):
        return '%20d' % int(value)
    if isinstance(value, (float, np.floating)):
        v = float(value)
        if v != v or v in (float('inf'), float('-inf')):
            raise ValueError('FITS cannot store %r' % v)
        s = repr(v).upper()
        if 'E' not in s and '.' not in s:
            s += '.0'
        return '%20s' % s
    if value is None:
        return ''
    s = str(value).replace("'", "''")
    return ("'%-8s'" % s).ljust(20)


def _ascii(text):
    """Header text is 7-bit ASCII: anything else becomes '?' (what astropy does when it reads such a card)."""
    return ''.join(ch if ' ' <= ch <= '~' else '?' for ch in text)


def _long_string_cards(head, value, comment):
    """A string that does not fit one card, in the CONTINUE convention astropy writes (and reads): the pieces end in
    '&', the following cards are `CONTINUE  'piece&'`, the comment rides on the last cards."""
    esc = str(value).replace("'", "''")
    pieces = []
    first = 80 - len(head) - 3                              # quote + piece + '&' + quote after the head: 67 for 'KEYWORD = ',
    if first < 1:                                           # less for a 'HIERARCH LONGKEY = ' head
        raise ValueError('keyword too long for a FITS card: %r' % head)
    while esc:
        n = first if not pieces else 67
        if len(esc) > n and esc[:n].endswith("'") and (len(esc[:n]) - len(esc[:n].rstrip("'"))) % 2 == 1:
            n -= 1                                          # never split an escaped quote pair
        pieces.append(esc[:n])
        esc = esc[n:]
    cards = []
    for i, pc in enumerate(pieces):
        last = i == len(pieces) - 1
        h = head if i == 0 else 'CONTINUE  '
        cards.append("%s'%s%s'" % (h, pc, '' if last else '&'))
    if comment:
        if len(cards[-1]) + 3 + len(comment) <= 80:
            cards[-1] += ' / ' + comment
        else:
            cards[-1] = cards[-1][:-1] + "&'" if len(cards[-1]) < 80 else cards[-1]
            if not cards[-1].endswith("&'"):                # full last piece: move its final character to a new card
                body = cards[-1][:-1]
                cards[-1] = body[:-1] + "&'"
                cards.append("CONTINUE  '%s&'" % body[-1])
            text = comment
            while text:
                chunk, text = text[:62], text[62:]
                cards.append("CONTINUE  '%s' / %s" % ('&' if text else '', chunk))
    assert all(len(c) <= 80 for c in cards), 'FITS card longer than 80 bytes'
    return ''.join(c.ljust(80) for c in cards)


def _format_card(card):
    if card.raw is not None:
        return card.raw
    key = card.key.upper()
    if key in _COMMENTARY:
        text = '' if card.value is None else _ascii(str(card.value))
        return ('%-8s%s' % (key, text))[:80].ljust(80)
    if len(key) > 8:
        head = 'HIERARCH %s = ' % key
    else:
        head = '%-8s= ' % key
    value = _ascii(card.value) if isinstance(card.value, str) else card.value
    comment = _ascii(card.comment) if card.comment else ''
    body = _format_value(value)
    img = head + body
    if isinstance(value, str) and len(img) > 80:
        return _long_string_cards(head, value, comment)     # e.g. DATAFILE = a full path (ApFindBadPixels.py:154)
    if comment:
        img += ' / ' + comment
    if len(img) > 80:
        img = img[:80]                                      # the value is intact; only the comment is cut
    return img.ljust(80)


class Header:
    """Ordered FITS header with a small dict-like API (subset of astropy.io.fits.Header)."""

    def __init__(self, cards=None):
        self.cards = list(cards) if cards else []

    # -- dict-like -------------------------------------------------------------------------------
    def _find(self, key):
        key = key.upper()
        for i, c in enumerate(self.cards):
            if c.key == key and key not in _COMMENTARY:
                return i
        return -1

    def __contains__(self, key):
        return self._find(key) >= 0

    def __getitem__(self, key):
        i = self._find(key)
        if i < 0:
            raise KeyError("Keyword '%s' not found." % key)
        return self.cards[i].value

    def get(self, key, default=None):
        i = self._find(key)
        return self.cards[i].value if i >= 0 else default

    def comment(self, key):
        i = self._find(key)
        return self.cards[i].comment if i >= 0 else ''

    def __setitem__(self, key, value):
        """hdr[key] = value or (value, comment); HISTORY/COMMENT always append (astropy semantics)."""
        key = key.upper()
        comment = None
        if isinstance(value, tuple):
            value, comment = (value + ('',))[:2]
        if key in _COMMENTARY:
            self.cards.append(Card(key, value, ''))
            return
        i = self._find(key)
        if i >= 0:
            c = self.cards[i]
            c.value = value
            if comment is not None:
                c.comment = comment
            c.raw = None
        else:
            self.cards.append(Card(key, value, comment or ''))

    def __delitem__(self, key):
        i = self._find(key)
        if i < 0:
            raise KeyError("Keyword '%s' not found." % key)
        del self.cards[i]

    def keys(self):
        return [c.key for c in self.cards]

    def items(self):
        return [(c.key, c.value) for c in self.cards]

    def history(self):
        return [c.value for c in self.cards if c.key == 'HISTORY']

    def copy(self):
        return Header([Card(c.key, c.value, c.comment, c.raw) for c in self.cards])

    # -- (de)serialisation -----------------------------------------------------------------------
    @classmethod
    def fromstring(cls, text):
        text = _ascii(text)                                 # a stray non-ASCII byte (e.g. a degree sign) reads as '?'
        cards = []
        for i in range(0, len(text), 80):
            img = text[i:i + 80]
            key = img[:8].rstrip().upper()
            if key == 'END':
                break
            if key == 'CONTINUE' and cards and isinstance(cards[-1].value, str) and cards[-1].value.endswith('&') \
                    and cards[-1].key not in _COMMENTARY:
                val, com = _parse_value(img[8:])            # long string (CONTINUE convention): glue the pieces
                prev = cards[-1]
                prev.value = prev.value[:-1] + (val if isinstance(val, str) else '')
                prev.comment = (prev.comment + ' ' + com).strip() if com else prev.comment
                prev.raw = (prev.raw or '') + img
                continue
            if key in _COMMENTARY:
                if img.strip() == '':
                    continue
                cards.append(Card(key, img[8:].rstrip(), '', img))
            elif img[8:10] == '= ':
                val, com = _parse_value(img[10:])
                cards.append(Card(key, val, com, img))
            elif img.startswith('HIERARCH') and '=' in img:
                k, rest = img[8:].split('=', 1)
                val, com = _parse_value(rest)
                cards.append(Card(k.strip().upper(), val, com, img))
            else:
                cards.append(Card(key, img[8:].rstrip(), '', img))
        return cls(cards)

    def tostring(self):
        out = ''.join(_format_card(c) for c in self.cards) + 'END'.ljust(80)
        pad = (-len(out)) % BLOCK
        return out + ' ' * pad


def _structural(hdr, data):
    """Header with the mandatory structural keywords rewritten for `data` (astropy does the same)."""
    dt = data.dtype
    drop = {'SIMPLE', 'BITPIX', 'NAXIS', 'EXTEND'} | {'NAXIS%d' % i for i in range(1, 10)}
    rest = [c for c in hdr.cards if c.key not in drop]
    bzero = None
    if dt == np.uint8:
        bitpix = 8
    elif dt == np.int16:
        bitpix = 16
    elif dt == np.uint16:
        bitpix, bzero = 16, 32768
    elif dt == np.int32:
        bitpix = 32
    elif dt == np.uint32:
        bitpix, bzero = 32, 2147483648
    elif dt == np.int64:
        bitpix = 64
    elif dt == np.float32:
        bitpix = -32
    elif dt == np.float64:
        bitpix = -64
    else:
        raise TypeError('cannot write dtype %s to FITS' % dt)
    head = [Card('SIMPLE', True, 'conforms to FITS standard'), Card('BITPIX', bitpix, 'array data type'),
            Card('NAXIS', data.ndim, 'number of array dimensions')]
    for i, n in enumerate(reversed(data.shape)):
        head.append(Card('NAXIS%d' % (i + 1), int(n), ''))
    had_extend = any(c.key == 'EXTEND' for c in hdr.cards)
    if had_extend:
        head.append(Card('EXTEND', True, ''))
    # the scaling keywords describe the SOURCE file's storage; the data written here are physical values, so they
    # always go and only the unsigned-integer convention puts its own pair back (astropy does the same)
    rest = [c for c in rest if c.key not in ('BSCALE', 'BZERO')]
    if bzero is not None:
        head += [Card('BSCALE', 1, ''), Card('BZERO', bzero, '')]
    return Header(head + rest)


def read(path, want_data=True):
    """Primary HDU -> (data, Header).  Mirrors fits.open(path, uint=True, do_not_scale_image_data=False)."""
    if not want_data:
        # the header blocks and whatever follows the data unit (extensions, kept verbatim) - not the data unit itself: the per-frame
        # flow reads the raw frame's header again for every output file (ApCalibrate.py:348-404), 34 MB each time in the first form
        hdr, pos = _read_header_only(path)
        naxis = int(hdr.get('NAXIS', 0))
        bitpix = int(hdr['BITPIX'])
        if bitpix not in _BITPIX_DTYPE:
            raise OSError('%s: unsupported BITPIX %d' % (path, bitpix))
        count = int(np.prod([int(hdr['NAXIS%d' % i]) for i in range(1, naxis + 1)])) if naxis > 0 else 0
        nbytes = count * abs(bitpix) // 8
        with open(path, 'rb') as f:
            f.seek(pos + ((nbytes + BLOCK - 1) // BLOCK) * BLOCK)
            hdr._tail = f.read()
        return None, hdr
    with open(path, 'rb') as f:
        raw = f.read()
    if raw[:6] != b'SIMPLE':
        raise OSError('%s is not a FITS file (no SIMPLE card).' % path)
    # header: whole 2880-byte blocks up to the END card
    pos = 0
    text = ''
    while True:
        block = raw[pos:pos + BLOCK]
        if len(block) < BLOCK:
            raise OSError('%s: header is truncated.' % path)
        pos += BLOCK
        s = block.decode('ascii', 'replace')
        text += s
        if any(s[i:i + 8] == 'END     ' for i in range(0, BLOCK, 80)):
            break
    hdr = Header.fromstring(text)
    naxis = int(hdr.get('NAXIS', 0))
    shape = tuple(int(hdr['NAXIS%d' % i]) for i in range(naxis, 0, -1))
    bitpix = int(hdr['BITPIX'])
    if bitpix not in _BITPIX_DTYPE:
        raise OSError('%s: unsupported BITPIX %d' % (path, bitpix))
    count = int(np.prod(shape)) if naxis > 0 else 0
    nbytes = count * abs(bitpix) // 8
    data = None
    if want_data and count > 0:
        if len(raw) < pos + nbytes:
            raise OSError('%s: data unit is truncated.' % path)
        arr = np.frombuffer(raw, dtype=_BITPIX_DTYPE[bitpix], count=count, offset=pos).reshape(shape)
        bscale = hdr.get('BSCALE', 1)
        bzero = hdr.get('BZERO', 0)
        if bitpix > 0 and bscale == 1 and bzero == 2 ** (bitpix - 1) and bitpix in (16, 32, 64):
            udt = {16: np.uint16, 32: np.uint32, 64: np.uint64}[bitpix]
            sdt = {16: np.int16, 32: np.int32, 64: np.int64}[bitpix]
            data = (arr.astype(sdt).view(udt) ^ udt(1 << (bitpix - 1))).astype(udt)
        elif bscale != 1 or bzero != 0:
            ft = np.float32 if (bitpix > 0 and bitpix <= 16) else np.float64
            data = (arr.astype(ft) * ft(bscale) + ft(bzero)).astype(ft)
        else:
            data = arr.astype(arr.dtype.newbyteorder('='))
    hdr._tail = raw[pos + ((nbytes + BLOCK - 1) // BLOCK) * BLOCK:]        # extensions, kept verbatim
    return data, hdr


def getheader(path):
    return read(path, want_data=False)[1]


def _disk_bytes(data):
    dt = data.dtype
    if dt == np.uint16:
        disk = (data ^ np.uint16(0x8000)).view(np.int16).astype('>i2')
    elif dt == np.uint32:
        disk = (data ^ np.uint32(0x80000000)).view(np.int32).astype('>i4')
    else:
        disk = data.astype(dt.newbyteorder('>'))
    payload = disk.tobytes()
    return payload + b'\0' * ((-len(payload)) % BLOCK)


def _extension_hdu(name, data, cards=None):
    """One IMAGE extension (what astropy's CCDData writer emits for the MASK / UNCERT planes)."""
    data = np.asarray(data)
    prim = _structural(Header(), data)
    head = [Card('XTENSION', 'IMAGE', 'Image extension')]
    head += [c for c in prim.cards if c.key in ('BITPIX', 'NAXIS') or c.key.startswith('NAXIS')]
    head += [Card('PCOUNT', 0, 'number of parameters'), Card('GCOUNT', 1, 'number of groups')]
    head += [c for c in prim.cards if c.key in ('BSCALE', 'BZERO')]
    head.append(Card('EXTNAME', name, 'extension name'))
    for k, v in (cards or {}).items():
        head.append(Card(k, *v) if isinstance(v, tuple) else Card(k, v))
    return Header(head).tostring().encode('ascii') + _disk_bytes(data)


def write(path, data, header=None, overwrite=True, extensions=None):
    """Writes a primary HDU, followed by `extensions` = [(extname, array, {key: value | (value, comment)}), ...]
    as IMAGE extensions, or else by the extensions `header` was read with, if any."""
    if os.path.exists(path) and not overwrite:
        raise OSError("File '%s' already exists." % path)
    data = np.asarray(data)
    src = header if header is not None else Header()
    if extensions and 'EXTEND' not in src:
        src = src.copy()
        src['EXTEND'] = True
    hdr = _structural(src, data)
    tail = getattr(header, '_tail', b'') if header is not None else b''
    if extensions:
        tail = b''.join(_extension_hdu(n, a, c) for n, a, c in extensions)
    tmp = str(path) + '.tmp%d' % os.getpid()
    with open(tmp, 'wb') as f:
        f.write(hdr.tostring().encode('ascii'))
        f.write(_disk_bytes(data))
        f.write(tail)
    os.replace(tmp, path)
    return hdr


def read_extension(path, extname):
    """(data, Header) of the first IMAGE extension called `extname` (case-insensitive)."""
    _, hdr = read(path, want_data=False)
    raw = hdr._tail
    pos = 0
    while pos < len(raw):
        text = ''
        start = pos
        while True:
            block = raw[pos:pos + BLOCK].decode('ascii', 'replace')
            if len(block) < BLOCK:
                raise OSError('%s: extension header is truncated.' % path)
            pos += BLOCK
            text += block
            if any(block[i:i + 8] == 'END     ' for i in range(0, BLOCK, 80)):
                break
        eh = Header.fromstring(text)
        naxis = int(eh.get('NAXIS', 0))
        shape = tuple(int(eh['NAXIS%d' % i]) for i in range(naxis, 0, -1))
        bitpix = int(eh['BITPIX'])
        count = int(np.prod(shape)) if naxis > 0 else 0
        nbytes = count * abs(bitpix) // 8 + int(eh.get('PCOUNT', 0))
        if str(eh.get('EXTNAME', '')).strip().upper() == extname.upper() and str(eh.get('XTENSION', '')).strip() == 'IMAGE':
            arr = np.frombuffer(raw, dtype=_BITPIX_DTYPE[bitpix], count=count, offset=pos).reshape(shape)
            if bitpix == 16 and eh.get('BZERO', 0) == 32768:
                return (arr.astype(np.int16).view(np.uint16) ^ np.uint16(0x8000)), eh
            return arr.astype(arr.dtype.newbyteorder('=')), eh
        pos += ((nbytes + BLOCK - 1) // BLOCK) * BLOCK
        if pos <= start:
            break
    raise KeyError("Extension '%s' not found in %s." % (extname, path))


# ---------------------------------------------------------------------------------------------------
# Device-side payload decode / encode (C ABI: apgpu_fits_decode / apgpu_fits_encode_f32).  The host
# only parses the header and moves raw bytes; the big-endian -> native conversion, the BZERO = 32768
# unsigned convention and the int16 -> float32 widening run as HIP kernels.
# ---------------------------------------------------------------------------------------------------
def _split_file(path):
    raw = np.fromfile(str(path), dtype=np.uint8)
    if raw.size < BLOCK or bytes(raw[:6]) != b'SIMPLE':
        raise OSError('%s is not a FITS file (no SIMPLE card).' % path)
    pos = 0
    text = ''
    while True:
        block = raw[pos:pos + BLOCK]
        if block.size < BLOCK:
            raise OSError('%s: header is truncated.' % path)
        pos += BLOCK
        t = block.tobytes().decode('ascii', 'replace')
        text += t
        if any(t[i:i + 8] == 'END     ' for i in range(0, BLOCK, 80)):
            break
    return raw, pos, Header.fromstring(text)


def read_device(path, device='cuda'):
    """Primary HDU -> (device tensor, Header).  uint16 for the BZERO = 32768 convention, float32 for plain
    int16 (widened like ApCalibrate._read_fits does) and for BITPIX -32; other layouts are decoded on the
    host by read() and uploaded."""
    import ctypes as C
    import torch
    from . import _lib
    raw, pos, hdr = _split_file(path)
    naxis = int(hdr.get('NAXIS', 0))
    shape = tuple(int(hdr['NAXIS%d' % i]) for i in range(naxis, 0, -1))
    bitpix = int(hdr['BITPIX'])
    count = int(np.prod(shape)) if naxis > 0 else 0
    nbytes = count * abs(bitpix) // 8
    bscale, bzero = hdr.get('BSCALE', 1), hdr.get('BZERO', 0)
    unsigned16 = bitpix == 16 and bscale == 1 and bzero == 32768
    plain = bscale == 1 and bzero == 0
    hdr._tail = raw[pos + ((nbytes + BLOCK - 1) // BLOCK) * BLOCK:].tobytes()
    if count == 0 or naxis != 2 or not (unsigned16 or (plain and bitpix in (16, -32, -64))):
        data, hdr2 = read(path)
        if data is None:
            return None, hdr2
        if data.dtype == np.uint16:
            t = torch.from_numpy(data.view(np.int16)).to(device).view(torch.uint16)
        else:
            t = torch.from_numpy(np.ascontiguousarray(data)).to(device)
        return t, hdr2
    if raw.size < pos + nbytes:
        raise OSError('%s: data unit is truncated.' % path)
    payload = torch.from_numpy(raw[pos:pos + nbytes].copy()).to(device)
    out = torch.empty(shape, dtype=torch.uint16 if unsigned16 else (torch.float64 if bitpix == -64 else torch.float32), device=device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.load().apgpu_fits_decode(C.c_void_p(payload.data_ptr()), bitpix, int(unsigned16),
                                            C.c_void_p(out.data_ptr()), count, stream))
    out._fits_payload = payload          # keep the source alive until the stream has consumed it
    return out, hdr


def _read_header_only(path):
    """(Header, byte offset of the data unit) of the primary HDU - reads the header blocks only."""
    text = ''
    pos = 0
    with open(str(path), 'rb') as f:
        while True:
            block = f.read(BLOCK)
            if pos == 0 and (len(block) < BLOCK or block[:6] != b'SIMPLE'):
                raise OSError('%s is not a FITS file (no SIMPLE card).' % path)
            if len(block) < BLOCK:
                raise OSError('%s: header is truncated.' % path)
            pos += BLOCK
            t = block.decode('ascii', 'replace')
            text += t
            if any(t[i:i + 8] == 'END     ' for i in range(0, BLOCK, 80)):
                break
    return Header.fromstring(text), pos


def _device_layout(hdr):
    """(shape, bitpix, unsigned16, nbytes) if apgpu_fits_decode handles the primary HDU, else None."""
    naxis = int(hdr.get('NAXIS', 0))
    if naxis != 2:
        return None
    shape = tuple(int(hdr['NAXIS%d' % i]) for i in range(naxis, 0, -1))
    bitpix = int(hdr['BITPIX'])
    bscale, bzero = hdr.get('BSCALE', 1), hdr.get('BZERO', 0)
    unsigned16 = bitpix == 16 and bscale == 1 and bzero == 32768
    plain = bscale == 1 and bzero == 0
    if not (unsigned16 or (plain and bitpix in (16, -32, -64))):
        return None
    return shape, bitpix, unsigned16, int(np.prod(shape)) * abs(bitpix) // 8


def _host_read_is_float64(h):
    """Whether read() returns float64 for this header: BITPIX -64, or BITPIX 32 / 64 integers scaled by a BSCALE / BZERO pair
    that is neither 1 / 0 nor the unsigned-integer convention (those stay integers)."""
    bitpix = int(h['BITPIX'])
    if bitpix == -64:
        return True
    if bitpix not in (32, 64):
        return False
    bscale, bzero = float(h.get('BSCALE', 1.0)), float(h.get('BZERO', 0.0))
    if bscale == 1.0 and bzero == 0.0:
        return False
    unsigned = bscale == 1.0 and bzero == float(2 ** (bitpix - 1))
    return not unsigned


def read_slab_device(paths, device='cuda', dtype='auto', timings=None):
    """N FITS files of one shape -> ONE contiguous [N, H, W] device slab + their Headers: the ingest of the stackers
    (the reference reads N files per combine: scripts/ap_combine_darks.py:411-420, core/ApCalibrate.py:260-328).

    The data units never exist as host arrays: a reader thread fills one of two PINNED staging buffers with the raw
    big-endian payload of file k + 1 while a copy stream uploads file k and apgpu_fits_decode byte-swaps / converts it
    straight into slab[k] (FITS payload -> device array, F1 of SURVEY 8(f)).  dtype 'auto': uint16 when every file uses
    the unsigned-16 convention (BITPIX 16, BZERO 32768), float64 when any file is BITPIX -64 (nothing is narrowed),
    float32 otherwise (plain int16 and uint16 widen exactly: core/ApCalibrate.py:304-307); or a torch float dtype.
    Layouts the decode kernel does not cover (scaled integers, BITPIX 8 / 32 / 64) are read on the host, per file.
    timings: a dict that receives wall times in seconds: 'headers', 'read' (disk -> pinned, reader thread), 'total'."""
    import ctypes as C
    import queue
    import threading
    import time
    import torch
    from . import _lib
    paths = [str(p) for p in paths]
    if not paths:
        raise ValueError('read_slab_device: no files')
    t_start = time.perf_counter()
    hdrs, offs, lays = [], [], []
    for p in paths:
        h, off = _read_header_only(p)
        hdrs.append(h)
        offs.append(off)
        lays.append(_device_layout(h))
    t_hdr = time.perf_counter()
    shapes = set()
    for p, h, lay in zip(paths, hdrs, lays):
        if int(h.get('NAXIS', 0)) != 2:
            raise RuntimeError('%s: expected a 2-D primary image, found NAXIS=%s.' % (p, h.get('NAXIS')))
        shapes.add(lay[0] if lay else tuple(int(h['NAXIS%d' % i]) for i in (2, 1)))
    if len(shapes) != 1:
        raise RuntimeError('Error, input images differ in shape: %s' % sorted(shapes))
    shape = shapes.pop()
    count = shape[0] * shape[1]
    if dtype == 'auto':
        if all(lay is not None and lay[2] for lay in lays):
            sdt = torch.uint16
        elif any((lay is not None and lay[1] == -64) or (lay is None and _host_read_is_float64(h))
                 for lay, h in zip(lays, hdrs)):
            # float64 for BITPIX -64 and for 32 / 64-bit integers with a real BSCALE / BZERO (read() - like astropy - hands
            # those over as float64, and the reference keeps floating data as it is); UNSCALED integers of any width go to
            # float32 like there (core/ApCalibrate.py:301-305 converts every non-float raw frame to float32)
            sdt = torch.float64
        else:
            sdt = torch.float32
    else:
        sdt = dtype
    dev = torch.device(device)
    slab = torch.empty((len(paths),) + shape, dtype=sdt, device=dev)
    lib = _lib.load()
    maxbytes = max([lay[3] for lay in lays if lay is not None] + [0])
    read_time = [0.0]
    if maxbytes:
        pinned = [torch.empty(maxbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        staged = [torch.empty(maxbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        free = [threading.Event(), threading.Event()]       # pinned buffer b may be overwritten
        for ev in free:
            ev.set()
        uploaded = [None, None]                             # CUDA event: the upload out of pinned buffer b has completed
        ready = queue.Queue(maxsize=2)

        def reader():
            try:
                b = 0
                for k, (p, lay) in enumerate(zip(paths, lays)):
                    if lay is None:
                        continue
                    free[b].wait()
                    if stop.is_set():
                        return
                    free[b].clear()
                    if uploaded[b] is not None:
                        uploaded[b].synchronize()
                    t0 = time.perf_counter()
                    view = memoryview(pinned[b].numpy())[:lay[3]]
                    with open(p, 'rb', buffering=0) as f:
                        f.seek(offs[k])
                        got = 0
                        while got < lay[3]:
                            n = f.readinto(view[got:])
                            if not n:
                                raise OSError('%s: data unit is truncated.' % p)
                            got += n
                    read_time[0] += time.perf_counter() - t0
                    ready.put((k, b, None))
                    b ^= 1
            except Exception as exc:                        # surfaces in the consumer
                ready.put((None, None, exc))
            ready.put((None, None, None))

        stop = threading.Event()
        th = threading.Thread(target=reader, daemon=True)
        th.start()
        cs = torch.cuda.Stream(device=dev)
        cs.wait_stream(torch.cuda.current_stream(dev))
        try:
          with torch.cuda.stream(cs):
            while True:
                k, b, exc = ready.get()
                if exc is not None:
                    raise exc
                if k is None:
                    break
                shp, bitpix, u16, nbytes = lays[k]
                staged[b][:nbytes].copy_(pinned[b][:nbytes], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(cs)
                uploaded[b] = ev
                free[b].set()
                ddt = torch.uint16 if u16 else (torch.float64 if bitpix == -64 else torch.float32)
                direct = ddt == sdt and slab[k].data_ptr() % 16 == 0          # the decode kernel uses 16-byte accesses
                dst = slab[k] if direct else torch.empty(shape, dtype=ddt, device=dev)
                _lib.check(lib.apgpu_fits_decode(C.c_void_p(staged[b].data_ptr()), bitpix, int(u16), C.c_void_p(dst.data_ptr()),
                                                 count, C.c_void_p(cs.cuda_stream)))
                if not direct:
                    if ddt == torch.uint16 and sdt != torch.uint16:      # widen exactly (torch has no uint16 arithmetic)
                        dst = dst.view(torch.int16).to(torch.int32) & 0xFFFF
                    if sdt == torch.uint16:
                        slab[k].view(torch.int16).copy_(dst.view(torch.int16))
                    else:
                        slab[k].copy_(dst.to(sdt))
        finally:
            # whatever happened above (a failed device call, an exception forwarded from the reader): release the reader -
            # it may be waiting for a staging buffer or for room in the queue - and wait for it, so that no thread and no
            # pinned buffer outlives the call
            stop.set()
            for ev in free:
                ev.set()
            while th.is_alive():
                try:
                    ready.get(timeout=0.05)
                except queue.Empty:
                    pass
            th.join()
        torch.cuda.current_stream(dev).wait_stream(cs)
        for t in staged:
            t.record_stream(torch.cuda.current_stream(dev))
    for k, (p, lay) in enumerate(zip(paths, lays)):         # layouts decoded on the host
        if lay is None:
            data, _ = read(p)
            if data.dtype == np.uint16:
                data = data.astype(np.int32)
            slab[k].copy_(torch.from_numpy(np.ascontiguousarray(data)).to(dev).to(sdt))
    if timings is not None:
        timings.update(headers=t_hdr - t_start, read=read_time[0], total=time.perf_counter() - t_start)
    return slab, hdrs


_staging_buf = {}


def _staging(nbytes):
    """A pinned host buffer of at least nbytes (one per thread, grown as needed): the D2H copy of a frame runs at the link's
    rate into pinned memory and at a fraction of it into pageable memory."""
    import threading
    import torch
    key = threading.get_ident()
    buf = _staging_buf.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, pin_memory=True)
        _staging_buf[key] = buf
    return buf[:nbytes]


class WritePool:
    """File writes of device frames overlapped with each other and with the caller's next frame (round 6): `workers` threads, each
    write from its own pinned staging buffer.  write_device(.., pool=p) returns once the frame's bytes are in a staging buffer (the
    device tensor may be reused); p.wait() returns when every file is in place and re-raises the first error.  Writing a 4096^2
    float32 frame into the page cache is a 67 MB copy on one core (~20 ms): four of them run side by side."""

    def __init__(self, workers=4):
        import queue
        from concurrent.futures import ThreadPoolExecutor
        self._ex = ThreadPoolExecutor(max_workers=workers)
        self._free = queue.Queue()
        for _ in range(workers):
            self._free.put([None])                           # one slot per worker; its buffer is allocated on first use
        self._futures = []

    def staging(self, nbytes):
        import torch
        slot = self._free.get()                              # blocks while every buffer is being written out
        if slot[0] is None or slot[0].numel() < nbytes:
            slot[0] = torch.empty(int(nbytes), dtype=torch.uint8, pin_memory=True)
        return slot

    def submit(self, slot, fn):
        def run():
            try:
                fn()
            finally:
                self._free.put(slot)
        self._futures.append(self._ex.submit(run))

    def wait(self):
        futures, self._futures = self._futures, []
        err = None
        for f in futures:
            try:
                f.result()
            except Exception as e:                           # noqa: BLE001 - reported after every write has finished
                err = err or e
        if err is not None:
            raise err

    def close(self):
        try:
            self.wait()
        finally:
            self._ex.shutdown(wait=True)


_shared_pool = []


def shared_write_pool(workers=None):
    """The process's WritePool (created on first use, its pinned buffers kept: pinning 67 MB costs more than writing it).
    APGPU_WRITE_THREADS sets the number of writer threads (default 4; one pinned frame-sized buffer each)."""
    if not _shared_pool:
        import atexit
        workers = int(workers or os.environ.get('APGPU_WRITE_THREADS', 4))
        _shared_pool.append(WritePool(max(1, workers)))
        atexit.register(_shared_pool[0].close)
    return _shared_pool[0]


def write_device(path, tensor, header=None, overwrite=True, pool=None):
    """float32 / float64 device tensor -> BITPIX -32 / -64 primary HDU (big-endian conversion on the device).
    pool: a WritePool - the file is written by one of its threads after this call returned (pool.wait() before reading it)."""
    import ctypes as C
    import torch
    from . import _lib
    if os.path.exists(path) and not overwrite:
        raise OSError("File '%s' already exists." % path)
    if tensor.dtype not in (torch.float32, torch.float64) or not tensor.is_cuda:
        return write(path, tensor.cpu().numpy() if tensor.dtype != torch.uint16
                     else tensor.view(torch.int16).cpu().numpy().view(np.uint16), header, overwrite)
    tensor = tensor.contiguous()
    f64 = tensor.dtype == torch.float64
    payload = torch.empty(tensor.numel() * (8 if f64 else 4), dtype=torch.uint8, device=tensor.device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    enc = _lib.load().apgpu_fits_encode_f64 if f64 else _lib.load().apgpu_fits_encode_f32
    _lib.check(enc(C.c_void_p(tensor.data_ptr()), C.c_void_p(payload.data_ptr()), tensor.numel(), stream))
    # device -> a pinned staging buffer (kept between calls) -> the file, without an intermediate bytes object: the pageable
    # .cpu() + .tobytes() of the first form were 25 of the 52 ms a 4096^2 float32 frame took to write (profiles/r06/frame_path_first.txt)
    nbytes = payload.numel()
    slot = pool.staging(nbytes) if pool is not None else None
    host = slot[0][:nbytes] if slot is not None else _staging(nbytes)
    host.copy_(payload, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    hdr = _structural(header if header is not None else Header(),
                      _ShapeOnly(tuple(tensor.shape), np.dtype(np.float64 if f64 else np.float32)))
    head = hdr.tostring().encode('ascii')
    tail = getattr(header, '_tail', b'') if header is not None else b''
    tmp = str(path) + '.tmp%d' % os.getpid()

    def put():
        data_bytes = memoryview(host.numpy())
        with open(tmp, 'wb') as f:
            f.write(head)
            f.write(data_bytes)
            f.write(b'\0' * ((-nbytes) % BLOCK))
            f.write(tail)
        os.replace(tmp, path)

    if pool is not None:
        pool.submit(slot, put)
    else:
        put()
    return hdr


class _ShapeOnly:
    """What _structural() needs from an array (dtype, ndim, shape) without materialising one."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype, self.ndim = shape, dtype, len(shape)
