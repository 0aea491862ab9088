"""FITS dump / load of the step outputs (reference steps.py:301-352 and
the lazy reload of DataObj.__get__, :131-160).

Golden files tests/golden/g9_*.fits were written by astropy.io.fits (oracle/gen_fits_golden.py)
in the layout mpdaf gives a cube without variance or mask: empty primary + IMAGE extension
'DATA'.  CPU tests pin the NumPy restatement (oracle/fits_ref.py) and the host header code
against them; GPU tests pin the device codec (csrc/fits.hip, through origin_amd/fitsio.py and
the C ABI) against both, bit for bit.
"""
import os
import subprocess

import numpy as np
import pytest

from oracle import fits_ref
from oracle import golden_cases as gc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FILES = {"cube64": "g9_cube_f64.fits", "cube32": "g9_cube_f32.fits", "prof8": "g9_cube_u8.fits",
         "area64": "g9_image_i64.fits", "img64": "g9_image_f64.fits"}
CONDA_PY = "/opt/conda/bin/python3.9"


def gpath(name):
    return os.path.join(GOLDEN, name)


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.fixture(scope="module")
def arrays():
    z = np.load(gpath("g9_arrays.npz"))
    return {k: z[k] for k in z.files}


# ------------------------------------------------------------------------------- CPU
def test_g9_inputs_are_reproducible(arrays):
    g = gc.g9_inputs()
    for k in FILES:
        assert same_bits(g[k], arrays[k]), k


@pytest.mark.parametrize("key", sorted(FILES))
def test_g9_oracle_matches_astropy_files(arrays, key):
    data, hdr = fits_ref.read_image(gpath(FILES[key]))
    assert same_bits(data, arrays[key])
    hdus = fits_ref.scan(gpath(FILES[key]))
    raw = open(gpath(FILES[key]), "rb").read()
    assert raw[hdus[1][1]:hdus[1][1] + hdus[1][2]] == fits_ref.encode(arrays[key], hdr["BITPIX"])
    assert len(raw) % 2880 == 0


def test_g9_table_oracle(arrays):
    cols = fits_ref.read_table(gpath("g9_table.fits"))
    for k in ("Tval_r", "Pval_r", "Det_m", "Det_M"):
        assert np.array_equal(cols[k], arrays["table_" + k])


def test_header_scan_agrees_with_oracle_on_astropy_files():
    from origin_amd import fitsio
    for name in list(FILES.values()) + ["g9_table.fits"]:
        ours, ref = fitsio.scan(gpath(name)), fits_ref.scan(gpath(name))
        assert len(ours) == len(ref) == 2
        for (h1, o1, n1), (h2, o2, n2) in zip(ours, ref):
            assert (o1, n1) == (o2, n2) and dict(h1) == dict(h2)


def test_header_cards_follow_the_fixed_format():
    from origin_amd import fitsio
    g = gc.g9_inputs()
    cards = [fitsio.card(k, v) for k, v in dict(g["wcs"], **g["wave"]).items()]
    cards += [fitsio.card("SIMPLE", True, "conforms"), fitsio.card("NAXIS", 3),
              fitsio.card("OBJECT", "it's"), fitsio.card("COMMENT", None, "free text")]
    for c in cards:
        assert len(c) == 80 and c.isascii()
        if c[8:10] == "= " and c[10] != "'":
            assert c[30:].strip() == "" or c[30:33] == " / "  # value ends at column 30
    hdr, end = fitsio.parse_header(fitsio.header_bytes(cards))
    assert end == 2880 and hdr["OBJECT"] == "it's" and hdr["NAXIS"] == 3 and hdr["SIMPLE"] is True
    for k, v in dict(g["wcs"], **g["wave"]).items():
        if isinstance(v, float):
            assert abs(hdr[k] - v) <= 1e-13 * abs(v)
        else:
            assert hdr[k] == v
    # the golden files carry the same cards: ours parse to the same values as astropy's
    gh = fitsio.scan(gpath("g9_cube_f64.fits"))[1][0]
    for k in dict(g["wcs"], **g["wave"]):
        assert gh[k] == hdr[k], k
    with pytest.raises(ValueError):
        fitsio.card("TOOLONGKEY", 1)
    with pytest.raises(ValueError):
        fitsio.card("X", float("nan"))


def test_table_writer_matches_astropy_layout(tmp_path, arrays):
    from origin_amd import fitsio
    cols = {k: arrays["table_" + k] for k in ("Tval_r", "Pval_r", "Det_m", "Det_M")}
    p = fitsio.write_table(str(tmp_path / "t.fits"), cols)
    ours, ref = fits_ref.scan(p), fits_ref.scan(gpath("g9_table.fits"))
    raw_o, raw_r = open(p, "rb").read(), open(gpath("g9_table.fits"), "rb").read()
    assert raw_o[ours[1][1]:] == raw_r[ref[1][1]:]          # data unit + padding, byte for byte
    for k in ("XTENSION", "BITPIX", "NAXIS", "NAXIS1", "NAXIS2", "PCOUNT", "GCOUNT", "TFIELDS",
              "TTYPE1", "TFORM1", "TTYPE4", "TFORM4"):
        assert ours[1][0][k] == ref[1][0][k], k
    back = fitsio.read_table(p)
    for k in cols:
        assert np.array_equal(back[k], cols[k]) and back[k].dtype == cols[k].dtype


# ------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def ctx():
    from origin_amd.device import default_context
    return default_context(0)


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(FILES))
def test_read_golden_files_on_device(ctx, arrays, key):
    from origin_amd import fitsio
    dev, hdr = fitsio.read_image(gpath(FILES[key]), ctx=ctx)
    want = arrays[key]
    got = dev.to_host()
    if want.dtype == np.int64:
        assert got.dtype == np.int32 and np.array_equal(got, want)
    else:
        assert same_bits(got, want)
    cube = fitsio.FitsCube(gpath(FILES[key]))
    assert cube.shape == want.shape and same_bits(cube._data, want)
    if want.dtype.kind == "f":   # the compute view: float32 on the device
        f32 = cube.device(ctx).to_host()
        with np.errstate(over="ignore"):
            assert same_bits(f32, want.astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(FILES))
def test_written_files_equal_astropy_bytes(ctx, arrays, tmp_path, key):
    """Same data unit (and padding) as astropy byte for byte; same mandatory cards."""
    from origin_amd import fitsio
    g = gc.g9_inputs()
    cards = dict(g["wcs"], **g["wave"]) if arrays[key].ndim == 3 else g["wcs"]
    p = fitsio.write_image(str(tmp_path / "o.fits"), arrays[key], header=cards, ctx=ctx)
    ours, ref = fits_ref.scan(p), fits_ref.scan(gpath(FILES[key]))
    raw_o, raw_r = open(p, "rb").read(), open(gpath(FILES[key]), "rb").read()
    assert len(raw_o) == len(raw_r)
    assert raw_o[ours[1][1]:] == raw_r[ref[1][1]:]
    for k, v in ref[1][0].items():
        assert ours[1][0][k] == v, k
    assert dict(ours[0][0]) == dict(ref[0][0])
    back, _ = fits_ref.read_image(p)
    assert same_bits(back, arrays[key])


@pytest.mark.gpu
def test_float32_device_cube_is_widened_like_the_reference(ctx, tmp_path):
    """A float32 cube in HBM tagged float64 (LazyCube) lands as BITPIX -64 holding exactly the
    float64 values of the float32 numbers; round trip at a size that takes several I/O chunks."""
    from origin_amd import fitsio
    from origin_amd.steps import LazyCube
    rng = np.random.default_rng(5)
    # 3 chunks of 64 MiB with a ragged tail on the float64 side
    x = rng.standard_normal((45, 700, 701)).astype(np.float32)
    x[0, 0, :3] = [np.inf, -0.0, 1e-45]
    dev = ctx.to_device(x)
    p = fitsio.write_image(str(tmp_path / "c.fits"), LazyCube(dev, dtype=np.float64), ctx=ctx)
    data, hdr = fits_ref.read_image(p)
    assert hdr["BITPIX"] == -64 and same_bits(data, x.astype(np.float64))
    back, _ = fitsio.read_image(p, dtype=np.float32, ctx=ctx)
    assert same_bits(back.to_host(), x)
    assert os.path.getsize(p) % 2880 == 0
    with pytest.raises(ValueError):   # a truncated file is reported, not read past its end
        with open(p, "r+b") as f:
            f.truncate(os.path.getsize(p) - 2880 * 4)
        fitsio.read_image(p, ctx=ctx)


@pytest.mark.gpu
def test_codec_round_trip_full_spectral_axis(ctx):
    """Size-independent property at the production channel count: decode(encode(x)) == x for
    every file type a step writes, straight through the C ABI."""
    import ctypes as C
    from origin_amd import _capi
    from origin_amd.device import DeviceArray
    rng = np.random.default_rng(11)
    n = 3681 * 64 * 64 + 3
    for src, code, bitpix in ((rng.standard_normal(n).astype(np.float32), 0, -64),
                              (rng.standard_normal(n).astype(np.float32), 0, -32),
                              (rng.integers(0, 256, n).astype(np.uint8), 1, 8),
                              (rng.integers(-2**31, 2**31, n).astype(np.int32), 2, 64),
                              (rng.integers(-2**31, 2**31, n).astype(np.int32), 2, 32),
                              (rng.standard_normal(n), 3, -64)):
        d = ctx.to_device(src)
        raw = DeviceArray(ctx, (n * abs(bitpix) // 8,), np.uint8)
        out = DeviceArray(ctx, (n,), src.dtype)
        _capi.call("origin_fits_encode", ctx.handle, d.p, code, n, bitpix, raw.p)
        _capi.call("origin_fits_decode", ctx.handle, raw.p, bitpix, n, code, out.p)
        assert same_bits(out.to_host(), src)
        # and the bytes are what the oracle says, on a slice the oracle converts in no time
        m = 100_003
        assert raw.to_host()[:m * abs(bitpix) // 8].tobytes() == fits_ref.encode(src[:m], bitpix)
        for a in (d, raw, out):
            a.free()
    with pytest.raises(_capi.OriginHipError):
        _capi.call("origin_fits_encode", ctx.handle, C.c_void_p(8), 0, 4, 24, C.c_void_p(8))


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no astropy interpreter on this box")
def test_astropy_reads_our_files(ctx, arrays, tmp_path):
    from origin_amd import fitsio
    g = gc.g9_inputs()
    paths = {}
    for key in FILES:
        cards = dict(g["wcs"], **g["wave"]) if arrays[key].ndim == 3 else g["wcs"]
        paths[key] = fitsio.write_image(str(tmp_path / f"{key}.fits"), arrays[key], header=cards,
                                        ctx=ctx)
    script = (
        "import sys, numpy as np\n"
        "for n, v in (('float', float), ('int', int), ('bool', bool), ('object', object),"
        " ('str', str), ('complex', complex)):\n"
        "    hasattr(np, n) or setattr(np, n, v)\n"
        "from astropy.io import fits\n"
        "for p in sys.argv[1:]:\n"
        "    with fits.open(p) as h:\n"
        "        h.verify('exception')\n"
        "        np.save(p + '.npy', h['DATA'].data)\n"
        "        print(p, h['DATA'].header['BITPIX'], h['DATA'].header.get('CRVAL1'))\n")
    r = subprocess.run([CONDA_PY, "-W", "ignore", "-c", script] + list(paths.values()),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for key, p in paths.items():
        got = np.load(p + ".npy")
        assert got.shape == arrays[key].shape
        assert got.astype(got.dtype.newbyteorder("=")).tobytes() == arrays[key].tobytes(), key


@pytest.mark.gpu
def test_step_dump_load_and_resume(ctx, tmp_path):
    """dump() after each step, load(), and the chain goes on from the files: same results as
    the chain that never left HBM (reference steps.py:301-352; status RUN -> DUMPED)."""
    from origin_amd import synth
    from origin_amd.fitsio import FitsCube
    from origin_amd.steps import SimpleOrig, Status

    f, raw, var, mask = synth.small_case(Nz=160, Ny=48, Nx=52, seed=3, psf_size=9, nprof=3)
    areamap = np.ones((48, 52), int)
    areamap[:, 26:] = 2

    def chain(dump):
        orig = SimpleOrig(raw, var, mask, f.PSF, f.profiles, ctx=ctx)
        out = str(tmp_path / ("dumped" if dump else "plain"))
        os.makedirs(out, exist_ok=True)
        for call in (lambda: orig.step01_preprocessing(),
                     lambda: orig.step02_areas.set_areamap(areamap),
                     lambda: orig.step03_compute_PCA_threshold(),
                     lambda: orig.step04_compute_greedy_PCA(),
                     lambda: orig.step05_compute_TGLR(),
                     lambda: orig.step06_compute_purity_threshold(purity=0.8)):
            call()
            if dump:
                for step in orig.steps.values():
                    step.dump(out)
                    step.load(out)
        return orig, out

    plain, _ = chain(False)
    dumped, out = chain(True)
    assert all(s.status is Status.DUMPED for s in dumped.steps.values())
    assert isinstance(dumped.steps["compute_TGLR"].__dict__["cube_correl"], (str, FitsCube))
    for name in ("cube_std", "cont_dct", "cube_faint", "cube_correl", "cube_correl_min",
                 "cube_profile", "cube_local_max", "cube_local_min"):
        a, b = getattr(dumped, name), getattr(plain, name)
        assert isinstance(a, FitsCube) and os.path.isfile(f"{out}/{name}.fits")
        assert same_bits(a._data, b._data), name
    assert dumped.cube_std._data.dtype == np.float64 and dumped.cont_dct._data.dtype == np.float32
    assert dumped.cube_profile._data.dtype == np.uint8
    for name in ("ima_std", "mapO2", "maxmap", "minmap", "segmap_merged", "areamap"):
        a = np.asarray(getattr(dumped, name)._data)
        b = np.asarray(getattr(getattr(plain, name), "_data", getattr(plain, name)))
        assert np.array_equal(a, b), name
    assert np.array_equal(dumped.thresO2, np.asarray(plain.thresO2))
    for c in ("Tval_r", "Pval_r", "Det_m", "Det_M"):
        assert np.array_equal(np.asarray(dumped.Pval[c], float), np.asarray(plain.Pval[c], float),
                              equal_nan=True), c
    assert dumped.param["threshold"] == plain.param["threshold"]
    # a step that was not run has nothing to dump
    fresh = SimpleOrig(raw, var, mask, f.PSF, f.profiles, ctx=ctx)
    fresh.steps["preprocessing"].dump(out)
    assert fresh.steps["preprocessing"].status is Status.NOTRUN


def test_scan_handles_headers_longer_than_one_block(tmp_path):
    """mpdaf / MUSE headers run over several 2880-byte blocks (hundreds of ESO HIERARCH cards)."""
    from origin_amd import fitsio
    extra = {f"K{i:03d}": float(i) for i in range(90)}          # 90 + mandatory cards: 3 blocks
    prim = fitsio.header_bytes(fitsio._image_cards((), 8, True, extra=extra))
    ext = fitsio.header_bytes(fitsio._image_cards((2, 3), -32, False, "DATA", extra))
    data = fits_ref.encode(np.arange(6, dtype=np.float32).reshape(2, 3), -32)
    p = tmp_path / "long.fits"
    p.write_bytes(prim + ext + data + b"\0" * (-len(data) % 2880))
    assert len(prim) == len(ext) == 3 * 2880
    hdus = fitsio.scan(str(p))
    assert len(hdus) == 2 and hdus[1][1] == 6 * 2880 and hdus[1][2] == 24
    assert hdus[1][0]["K089"] == 89.0 and hdus[1][0]["EXTNAME"] == "DATA"
    assert [dict(h[0]) for h in hdus] == [dict(h[0]) for h in fits_ref.scan(str(p))]
    assert fitsio.find_hdu(str(p))[1] == 6 * 2880
