"""CPU: liborigin_hip.so builds for gfx950, loads, and exports every function that
include/origin_hip.h declares (no compute call is made without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "origin_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(origin_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from origin_amd import build, _capi
    build.build()
    return _capi.load()


def test_library_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert len(names) >= 30
    raw = ctypes.CDLL(os.path.join(ROOT, "origin_amd", "liborigin_hip.so"))
    for n in names:
        assert hasattr(raw, n), f"{n} declared in origin_hip.h but not exported"


def test_binding_covers_the_header(lib):
    from origin_amd import _capi
    assert sorted(_capi.SIGNATURES) == declared_functions()


def test_abi_version_and_error_string(lib):
    assert lib.origin_abi_version() == 1
    assert isinstance(lib.origin_last_error(), bytes)


def test_no_gpu_fails_loudly(lib):
    """Without a device the product raises; it never falls back to the CPU."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    from origin_amd import _capi
    from origin_amd.device import Context
    with pytest.raises(_capi.OriginHipError) as e:
        Context(0)
    assert e.value.code in (-4, -3)
    import numpy as np
    import origin_amd.lib_origin as hip
    with pytest.raises(RuntimeError):
        hip.O2test(np.zeros((4, 3), np.float32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "origin_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+\.*oracle", src, flags=re.M), \
                f"{fn} imports the oracle"
