"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the FITS rules the dump / load row needs
(SURVEY 8f-4): header blocks of 80-character cards, data units as big-endian arrays padded
to 2880 bytes (FITS standard 4.0, sections 3-5, 7.1, 7.3).  NumPy only.

It restates what astropy.io.fits does underneath mpdaf's ``Cube.write`` / ``Cube(path)``
(reference call sites steps.py:141-146, :319); mpdaf is absent from the reference tree, so
this oracle is pinned against files written by astropy itself (oracle/gen_fits_golden.py ->
tests/golden/g9_*.fits), not against mpdaf: "parity unpinned" with respect to mpdaf's exact
header contents.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it; the product path (origin_amd/fitsio.py + csrc/fits.hip) never does.
"""
from collections import OrderedDict

import numpy as np

BLOCK, CARD = 2880, 80
DTYPE = {-64: ">f8", -32: ">f4", 8: "u1", 16: ">i2", 32: ">i4", 64: ">i8"}


def encode(array, bitpix):
    """Bytes of the data unit (without padding) of ``array`` stored with ``bitpix``."""
    return np.ascontiguousarray(array).astype(DTYPE[bitpix]).tobytes()


def decode(raw, bitpix, shape):
    """Native-endian array of a data unit."""
    a = np.frombuffer(raw, dtype=DTYPE[bitpix]).reshape(shape)
    return a.astype(a.dtype.newbyteorder("="))


def _value(field):
    s = field.strip()
    if not s:
        return None
    if s.startswith("'"):
        end = 1
        while True:
            end = s.index("'", end)
            if s[end:end + 2] == "''":
                end += 2
                continue
            break
        return s[1:end].replace("''", "'").rstrip()
    s = s.split("/")[0].strip()
    if s in ("T", "F"):
        return s == "T"
    try:
        return int(s)
    except ValueError:
        return float(s.replace("D", "E"))


def scan(path):
    """[(header OrderedDict, data offset, data bytes)] per HDU."""
    raw = open(path, "rb").read()
    out, pos = [], 0
    while pos < len(raw):
        hdr = OrderedDict()
        done = False
        while not done:
            block = raw[pos:pos + BLOCK].decode("ascii")
            pos += BLOCK
            for i in range(0, BLOCK, CARD):
                c = block[i:i + CARD]
                if c[:8].strip() == "END":
                    done = True
                    break
                if c[8:10] == "= ":
                    hdr[c[:8].strip()] = _value(c[10:])
        n = 0
        if hdr.get("NAXIS", 0):
            n = 1
            for i in range(1, hdr["NAXIS"] + 1):
                n *= hdr[f"NAXIS{i}"]
            n = abs(hdr["BITPIX"]) // 8 * hdr.get("GCOUNT", 1) * (hdr.get("PCOUNT", 0) + n)
        out.append((hdr, pos, n))
        pos += n + (-n % BLOCK)
    return out


def read_image(path, ext="DATA"):
    for hdr, off, nb in scan(path):
        if hdr.get("EXTNAME") == ext:
            shape = tuple(hdr[f"NAXIS{i}"] for i in range(hdr["NAXIS"], 0, -1))
            raw = open(path, "rb").read()[off:off + nb]
            return decode(raw, hdr["BITPIX"], shape), hdr
    raise KeyError(ext)


def read_table(path):
    for hdr, off, nb in scan(path):
        if hdr.get("XTENSION") == "BINTABLE":
            form = {"D": ">f8", "K": ">i8", "J": ">i4", "E": ">f4"}
            dt = [(hdr[f"TTYPE{i}"], form[str(hdr[f"TFORM{i}"]).strip().lstrip("1")])
                  for i in range(1, hdr["TFIELDS"] + 1)]
            rec = np.frombuffer(open(path, "rb").read()[off:off + nb], dtype=dt)
            return OrderedDict((n, rec[n].astype(rec[n].dtype.newbyteorder("="))) for n, _ in dt)
    raise KeyError("BINTABLE")
