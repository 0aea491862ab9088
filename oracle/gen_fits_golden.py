"""TEST INFRASTRUCTURE ONLY -- FITS golden files for the dump / load row.

    /opt/conda/bin/python3.9 oracle/gen_fits_golden.py

The reference writes its step outputs through mpdaf (steps.py:301-340), which is not part of
the reference tree or of this image, so the files cannot come from the reference itself
("parity unpinned" against mpdaf).  They are written here by astropy.io.fits 4.3.1 -- the
library mpdaf writes through -- in mpdaf's layout for an object without variance or mask:
header-only primary HDU + IMAGE extension 'DATA'.  oracle/fits_ref.py (NumPy restatement of
the FITS data-unit rules) is checked against them on the spot.
"""
import os
import sys

import numpy as np

# aliases removed in numpy 1.24 that astropy 4.3.1 still touches at import
for _n, _v in (("float", float), ("int", int), ("bool", bool), ("object", object),
               ("str", str), ("complex", complex)):
    if not hasattr(np, _n):
        setattr(np, _n, _v)
if not hasattr(np, "asscalar"):
    np.asscalar = lambda a: a.item()
if not hasattr(np, "alen"):
    np.alen = len

from astropy.io import fits  # noqa: E402
from astropy.table import Table  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import fits_ref, golden_cases as gc  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def image_file(name, data, cards):
    hdr = fits.Header()
    for k, v in cards.items():
        hdr[k] = v
    hdul = fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=data, header=hdr, name="DATA")])
    path = os.path.join(OUT, name)
    hdul.writeto(path, overwrite=True)
    # the restatement must produce astropy's data unit byte for byte
    hdus = fits_ref.scan(path)
    h, off, nb = hdus[1]
    raw = open(path, "rb").read()[off:off + nb]
    assert raw == fits_ref.encode(data, h["BITPIX"]), name
    back = fits_ref.decode(raw, h["BITPIX"], data.shape)
    assert back.tobytes() == np.ascontiguousarray(data).tobytes(), name
    for k, v in cards.items():  # astropy keeps float cards within 20 characters (~14 digits)
        ok = abs(h[k] - v) <= 1e-13 * abs(v) if isinstance(v, float) else h[k] == v
        assert ok, (name, k, h[k], v)
    print("wrote", path, "BITPIX", h["BITPIX"], nb, "bytes: restatement agrees")


def main():
    g = gc.g9_inputs()
    cube_cards = dict(g["wcs"], **g["wave"])
    image_file("g9_cube_f64.fits", g["cube64"], cube_cards)
    image_file("g9_cube_f32.fits", g["cube32"], cube_cards)
    image_file("g9_cube_u8.fits", g["prof8"], cube_cards)
    image_file("g9_image_i64.fits", g["area64"], g["wcs"])
    image_file("g9_image_f64.fits", g["img64"], g["wcs"])
    t = Table([g["table"][k] for k in ("Tval_r", "Pval_r", "Det_m", "Det_M")],
              names=("Tval_r", "Pval_r", "Det_m", "Det_M"))
    path = os.path.join(OUT, "g9_table.fits")
    t.write(path, overwrite=True)
    cols = fits_ref.read_table(path)
    for k in g["table"]:
        assert np.array_equal(cols[k], g["table"][k]), k
    print("wrote", path)
    np.savez(os.path.join(OUT, "g9_arrays.npz"), cube64=g["cube64"], cube32=g["cube32"],
             prof8=g["prof8"], area64=g["area64"], img64=g["img64"],
             **{"table_" + k: v for k, v in g["table"].items()})


if __name__ == "__main__":
    main()
