"""FITS files of the step outputs: what ``Step.dump`` / ``Step.load``
exchange with the disk (reference muse_origin/steps.py:301-352, lazy reload :131-160).

The reference writes every cube / image through mpdaf (``obj.write(outf,
convert_float32=False)``, steps.py:319): a header-only primary HDU followed by an IMAGE
extension named ``DATA`` holding the array (float64 unless the array is float32 / integer),
and reads it back with ``Cube(path)`` / ``Image(path)``, which take the ``DATA`` extension.
mpdaf is not part of the reference tree (nor of this image), so that layout is restated from
the FITS standard and mpdaf's documented behaviour and pinned against ``astropy.io.fits``
(the library mpdaf itself writes through): tests/golden/g9_*.fits were written by astropy,
and files written here are read back by astropy where it is installed.

Headers are host work (a few 80-character cards).  The data unit -- the array widened to
the file type and byte-swapped to big-endian -- is produced and consumed on the GPU
(csrc/fits.hip: ``origin_fits_write_data`` / ``origin_fits_read_data`` stream it to / from
the file descriptor in 64 MiB chunks through pinned buffers, conversion, PCIe copy and file
I/O overlapping), so a cube in HBM reaches the file without a host-side conversion pass.  No CPU fallback: without the
library these functions raise.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np

from . import _capi
from .device import DeviceArray, default_context

BLOCK = 2880
CARD = 80

_TYPE_CODE = {np.dtype(np.float32): 0, np.dtype(np.uint8): 1, np.dtype(np.int32): 2,
              np.dtype(np.float64): 3}
_BITPIX_OF = {np.dtype(np.float64): -64, np.dtype(np.float32): -32, np.dtype(np.uint8): 8,
              np.dtype(np.int16): 16, np.dtype(np.int32): 32, np.dtype(np.int64): 64,
              np.dtype(bool): 8}
_DTYPE_OF = {-64: np.dtype(np.float64), -32: np.dtype(np.float32), 8: np.dtype(np.uint8),
             16: np.dtype(np.int16), 32: np.dtype(np.int32), 64: np.dtype(np.int64)}
# element type a data unit decodes to on the device when the caller wants "what the file holds"
_NATIVE_DEV = {-64: np.dtype(np.float64), -32: np.dtype(np.float32), 8: np.dtype(np.uint8),
               16: np.dtype(np.int32), 32: np.dtype(np.int32), 64: np.dtype(np.int32)}


# ------------------------------------------------------------------------------- headers
def _format_value(value):
    if isinstance(value, (bool, np.bool_)):
        return "%20s" % ("T" if value else "F")
    if isinstance(value, (int, np.integer)):
        return "%20d" % int(value)
    if isinstance(value, (float, np.floating)):
        v = float(value)
        if not np.isfinite(v):
            raise ValueError("FITS header values must be finite")
        s = "%.16G" % v
        if "E" in s:
            m, e = s.split("E")
            if "." not in m:
                m += ".0"
            # the value field of a fixed-format card is 20 characters: drop trailing digits of
            # the significand (as astropy.io.fits does) rather than widen the field
            e = "E%+03d" % int(e)
            s = m[:max(20 - len(e), 3)] + e
        else:
            if "." not in s:
                s += ".0"
            s = s[:20]
        return "%20s" % s
    if isinstance(value, str):
        s = value.replace("'", "''")
        if len(s) > 68:
            raise ValueError("string values longer than 68 characters are not supported")
        return "'%-8s'" % s
    raise TypeError(f"unsupported header value {value!r}")


def card(key, value=None, comment=None):
    """One 80-character header card (FITS standard 4.0, fixed format)."""
    key = key.upper()
    if len(key) > 8 or not all(c.isalnum() or c in "-_" for c in key):
        raise ValueError(f"bad FITS keyword {key!r}")
    if key in ("COMMENT", "HISTORY", "") or value is None:
        text = "%-8s%s" % (key, (" " + str(comment)) if comment else "")
        return text[:CARD].ljust(CARD)
    text = "%-8s= %s" % (key, _format_value(value))
    if comment:
        text += " / " + comment
    return text[:CARD].ljust(CARD)


def header_bytes(cards):
    """Cards + END, padded with blanks to a multiple of 2880 bytes."""
    text = "".join(cards) + "END".ljust(CARD)
    text += " " * (-len(text) % BLOCK)
    return text.encode("ascii")


def _parse_value(field):
    s = field.strip()
    if not s:
        return None
    if s[0] == "'":
        out, i = [], 1
        while i < len(s):
            if s[i] == "'":
                if i + 1 < len(s) and s[i + 1] == "'":
                    out.append("'")
                    i += 2
                    continue
                break
            out.append(s[i])
            i += 1
        return "".join(out).rstrip()
    s = s.split("/")[0].strip()
    if s == "T":
        return True
    if s == "F":
        return False
    try:
        return int(s)
    except ValueError:
        return float(s.replace("D", "E"))


def parse_header(buf, offset=0):
    """(OrderedDict keyword -> value, offset of the byte after the header's last block)."""
    hdr = OrderedDict()
    pos = offset
    while True:
        block = bytes(buf[pos:pos + BLOCK])
        if len(block) < BLOCK:
            raise ValueError("truncated FITS header")
        pos += BLOCK
        for i in range(0, BLOCK, CARD):
            c = block[i:i + CARD].decode("ascii")
            key = c[:8].strip()
            if key == "END":
                return hdr, pos
            if c[8:10] == "= ":
                hdr[key] = _parse_value(c[10:])


def data_bytes(hdr):
    """Size of the data unit a header announces (without the padding)."""
    naxis = hdr.get("NAXIS", 0)
    if naxis == 0:
        return 0
    n = 1
    for i in range(1, naxis + 1):
        n *= hdr[f"NAXIS{i}"]
    return (abs(hdr["BITPIX"]) // 8) * hdr.get("GCOUNT", 1) * (hdr.get("PCOUNT", 0) + n)


def scan(path):
    """[(header, data offset, data bytes)] for every HDU of a file (headers only are read)."""
    out = []
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        pos = 0
        while pos < size:
            f.seek(pos)
            buf = f.read(BLOCK)
            # headers longer than one block: keep reading until END
            while True:
                try:
                    hdr, end = parse_header(buf)
                    break
                except ValueError:
                    more = f.read(BLOCK)
                    if not more:
                        raise
                    buf += more
            nb = data_bytes(hdr)
            out.append((hdr, pos + end, nb))
            pos += end + nb + (-nb % BLOCK)
    return out


# ------------------------------------------------------------------------------- writing
def _device_source(ctx, data):
    """(DeviceArray, host dtype the reference would hold) for anything a DataObj may hold."""
    dev = getattr(data, "dev", None)
    if dev is not None:  # steps.LazyCube
        if hasattr(dev, "gathered"):   # in pieces on several devices (session.TiledCube)
            return dev.gathered(ctx), np.dtype(getattr(data, "_dtype", dev.dtype))
        return dev, np.dtype(getattr(data, "_dtype", dev.dtype))
    if isinstance(data, DeviceArray):
        return data, data.dtype
    host = np.asarray(getattr(data, "_data", data))
    ref_dtype = host.dtype
    if host.dtype == bool:
        host = host.astype(np.uint8)
    elif host.dtype in (np.dtype(np.int64), np.dtype(np.int16)):
        if host.size and (host.max() > np.iinfo(np.int32).max or host.min() < np.iinfo(np.int32).min):
            raise ValueError("integer images beyond int32 are not supported")
        host = host.astype(np.int32)
    elif host.dtype not in _TYPE_CODE:
        host = host.astype(np.float64)
        ref_dtype = host.dtype
    return ctx.to_device(host), np.dtype(ref_dtype)


def _image_cards(shape, bitpix, primary, name=None, extra=None):
    cards = [card("SIMPLE", True, "conforms to FITS standard") if primary
             else card("XTENSION", "IMAGE", "Image extension"),
             card("BITPIX", bitpix, "array data type"),
             card("NAXIS", len(shape), "number of array dimensions")]
    for i, n in enumerate(reversed(shape)):
        cards.append(card(f"NAXIS{i + 1}", int(n)))
    if primary:
        cards.append(card("EXTEND", True))
    else:
        cards += [card("PCOUNT", 0, "number of parameters"), card("GCOUNT", 1, "number of groups")]
        if name:
            cards.append(card("EXTNAME", name, "extension name"))
    reserved = {"SIMPLE", "XTENSION", "BITPIX", "NAXIS", "EXTEND", "PCOUNT", "GCOUNT", "EXTNAME",
                "END"}
    for k, v in (extra or {}).items():
        if k.upper() in reserved or k.upper().startswith("NAXIS"):
            continue
        cards.append(card(k, *(v if isinstance(v, tuple) else (v,))))
    return cards


def _write_data_unit(f, ctx, dev, bitpix):
    """Append the data unit of a device array to the (unbuffered) open file, padded to a block.
    Conversion, PCIe copy and write() overlap inside the library (origin_fits_write_data)."""
    n = dev.size
    _capi.call("origin_fits_write_data", ctx.handle, dev.p, _TYPE_CODE[dev.dtype], n, bitpix,
               f.fileno())
    f.write(b"\0" * (-(n * (abs(bitpix) // 8)) % BLOCK))


def write_image(path, data, name="DATA", header=None, bitpix=None, ctx=None,
                primary_header=None):
    """Write one array as ``<primary, no data> + <IMAGE extension `name`>`` -- the layout of
    mpdaf's ``DataArray.write`` for an object without variance or mask, which is what
    ``Step.store_cube`` / ``store_image`` create (steps.py:284-299, mask=nomask).

    ``data``: DeviceArray, steps.LazyCube or anything array-like (uploaded).  ``bitpix``
    defaults to the FITS type of the dtype the reference would hold (float64 -> -64 as with
    ``convert_float32=False``)."""
    ctx = ctx or default_context(0)
    dev, ref_dtype = _device_source(ctx, data)
    if bitpix is None:
        bitpix = _BITPIX_OF.get(ref_dtype, -64)
    if dev.dtype not in _TYPE_CODE:
        raise TypeError(f"no device codec for {dev.dtype}")
    tmp = path + ".part"
    with open(tmp, "wb", buffering=0) as f:
        f.write(header_bytes(_image_cards((), 8, True, extra=primary_header)))
        f.write(header_bytes(_image_cards(dev.shape, bitpix, False, name, header)))
        if dev.size:
            _write_data_unit(f, ctx, dev, bitpix)
    os.replace(tmp, path)
    return path


# ------------------------------------------------------------------------------- reading
def find_hdu(path, ext="DATA"):
    """(header, data offset, data bytes) of the extension called ``ext`` (or number ``ext``);
    like mpdaf, falls back to the first HDU with data when no ``DATA`` extension exists."""
    hdus = scan(path)
    if isinstance(ext, int):
        return hdus[ext]
    for h in hdus:
        if str(h[0].get("EXTNAME", "")).upper() == ext.upper():
            return h
    for h in hdus:
        if h[2] and h[0].get("XTENSION", "IMAGE") == "IMAGE":
            return h
    raise KeyError(f"no image data in {path}")


def read_image(path, ext="DATA", dtype=None, ctx=None):
    """(DeviceArray, header) of an image HDU, decoded on the device.  ``dtype``: element type
    of the device array (float32 for the compute kernels); None keeps what the file holds
    (BITPIX -64 -> float64, -32 -> float32, 8 -> uint8, 16/32/64 -> int32)."""
    ctx = ctx or default_context(0)
    hdr, off, nb = find_hdu(path, ext)
    bitpix = hdr["BITPIX"]
    if hdr.get("BSCALE", 1) != 1 or hdr.get("BZERO", 0) != 0:
        raise ValueError("scaled images (BSCALE / BZERO) are not supported")
    shape = tuple(hdr[f"NAXIS{i}"] for i in range(hdr["NAXIS"], 0, -1))
    dtype = np.dtype(dtype) if dtype is not None else _NATIVE_DEV[bitpix]
    if dtype not in _TYPE_CODE:
        raise TypeError(f"no device codec for {dtype}")
    out = DeviceArray(ctx, shape, dtype)
    n, width = out.size, abs(bitpix) // 8
    if os.path.getsize(path) < off + nb:
        raise ValueError(f"{path}: truncated data unit")
    if n * width != nb:
        raise ValueError(f"{path}: header announces {nb} bytes for {n} elements")
    if n:
        with open(path, "rb", buffering=0) as f:
            f.seek(off)
            _capi.call("origin_fits_read_data", ctx.handle, f.fileno(), bitpix, n,
                       _TYPE_CODE[dtype], out.p)
    return out, hdr


class FitsCube:
    """A dumped cube / image reloaded on demand (what ``DataObj.__get__`` of the reference
    turns a path into, steps.py:141-146): ``._data`` is the host array with the file's dtype,
    ``.device(ctx)`` the float32 (or given) device array, both decoded from the file by the
    GPU and cached."""

    def __init__(self, path, ext="DATA"):
        self.path, self.ext = path, ext
        self._host = None
        self._dev = {}
        hdr, _, _ = find_hdu(path, ext)
        self.header = hdr
        self.shape = tuple(hdr[f"NAXIS{i}"] for i in range(hdr["NAXIS"], 0, -1))
        self._dtype = _DTYPE_OF[hdr["BITPIX"]]

    @property
    def _data(self):
        if self._host is None:
            dev, _ = read_image(self.path, self.ext, None)
            self._host = dev.to_host().astype(self._dtype, copy=False)
            dev.free()
        return self._host

    data = _data

    @property
    def dev(self):
        return None

    def device(self, ctx, dtype=np.float32):
        key = np.dtype(dtype)
        if key not in self._dev:
            self._dev[key] = read_image(self.path, self.ext, key, ctx)[0]
        return self._dev[key]


# ------------------------------------------------------------------------------- tables
def write_table(path, columns):
    """Binary-table extension with float64 ('D') / int64 ('K') columns from a mapping
    name -> 1-D array (the purity tables of step 6, steps.py:853-854, a few dozen rows: host
    work).  Layout as astropy's ``Table.write(format='fits')``: empty primary + BINTABLE."""
    names = list(columns)
    cols = [np.asarray(columns[k]) for k in names]
    nrows = len(cols[0]) if cols else 0
    fields = []
    for c in cols:
        if c.ndim != 1 or len(c) != nrows:
            raise ValueError("table columns must be 1-D and of equal length")
        fields.append((">i8", "K") if np.issubdtype(c.dtype, np.integer) or c.dtype == bool
                      else (">f8", "D"))
    rec = np.zeros(nrows, dtype=[(n, f[0]) for n, f in zip(names, fields)])
    for n, c in zip(names, cols):
        rec[n] = c
    cards = [card("XTENSION", "BINTABLE", "binary table extension"), card("BITPIX", 8),
             card("NAXIS", 2), card("NAXIS1", rec.dtype.itemsize), card("NAXIS2", nrows),
             card("PCOUNT", 0), card("GCOUNT", 1), card("TFIELDS", len(names))]
    for i, (n, f) in enumerate(zip(names, fields)):
        cards += [card(f"TTYPE{i + 1}", n), card(f"TFORM{i + 1}", f[1])]
    raw = rec.tobytes()
    tmp = path + ".part"
    with open(tmp, "wb") as f:
        f.write(header_bytes(_image_cards((), 8, True)))
        f.write(header_bytes(cards))
        f.write(raw + b"\0" * (-len(raw) % BLOCK))
    os.replace(tmp, path)
    return path


def read_table(path):
    """OrderedDict name -> array of the first BINTABLE extension ('D', 'K', 'J', 'E' columns)."""
    for hdr, off, nb in scan(path):
        if hdr.get("XTENSION") == "BINTABLE":
            break
    else:
        raise KeyError(f"no binary table in {path}")
    form = {"D": ">f8", "K": ">i8", "J": ">i4", "E": ">f4", "L": "u1"}
    dt = []
    for i in range(1, hdr["TFIELDS"] + 1):
        tform = str(hdr[f"TFORM{i}"]).strip().lstrip("1")
        if tform not in form:
            raise ValueError(f"unsupported column format {hdr[f'TFORM{i}']!r}")
        dt.append((str(hdr[f"TTYPE{i}"]), form[tform]))
    with open(path, "rb") as f:
        f.seek(off)
        rec = np.frombuffer(f.read(hdr["NAXIS1"] * hdr["NAXIS2"]), dtype=dt)
    return OrderedDict((n, rec[n].astype(rec[n].dtype.newbyteorder("="))) for n, _ in dt)
