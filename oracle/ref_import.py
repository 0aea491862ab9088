"""TEST INFRASTRUCTURE ONLY -- never imported by the product (origin_amd/).

Loader for the *real* reference numerics (``/root/reference/muse_origin/lib_origin.py``)
so that golden vectors can be generated from the reference's own functions
(SURVEY.md section 8c).  Only usable in the build container, with
``/opt/conda/bin/python3.9`` (numpy 1.26 / scipy 1.7 / astropy 4.3 / joblib).

``import muse_origin`` fails in this checkout for ordinary reasons (no generated
``version.py``, no ``mpdaf``, no ``photutils``).  ``lib_origin.py`` itself only
needs three names that carry no hot-path arithmetic:

* ``mpdaf.tools.progressbar`` (a tqdm wrapper),
* ``mpdaf.obj.Image`` (used by ``compute_deblended_segmap`` only),
* ``photutils.detect_sources`` (used by ``source_masks`` only),

so this module materialises, in a temporary directory OUTSIDE the repository,

    <tmp>/muse_origin/__init__.py          (empty apart from numpy alias patch)
    <tmp>/muse_origin/lib_origin.py   ->   symlink to the reference file
    <tmp>/muse_origin/source_masks.py ->   symlink to the reference file
    <tmp>/mpdaf/{__init__,obj,tools}.py    (inert stand-ins)
    <tmp>/photutils/__init__.py            (inert stand-in)

and puts it first on ``sys.path`` (on disk, not only in ``sys.modules``, so
joblib's loky workers can re-import it).  The reference source is never copied.
"""
import os
import sys
import tempfile

REFERENCE = os.environ.get("ORIGIN_REFERENCE", "/root/reference")

_INIT = '''\
import numpy as _np
# aliases removed in numpy 1.24 that astropy 4.3.1 still touches at import
for _n, _v in (("float", float), ("int", int), ("bool", bool), ("object", object),
               ("str", str), ("complex", complex)):
    if not hasattr(_np, _n):
        setattr(_np, _n, _v)
if not hasattr(_np, "asscalar"):
    _np.asscalar = lambda a: a.item()
if not hasattr(_np, "alen"):
    _np.alen = len
'''

_MPDAF_TOOLS = '''\
class progressbar:
    """Inert stand-in for mpdaf.tools.progressbar (tqdm wrapper)."""
    def __init__(self, iterable=None, total=None, **kw):
        self.iterable = iterable
        self.n = 0
    def __iter__(self):
        return iter(self.iterable)
    def __enter__(self):
        return self
    def __exit__(self, *a):
        return False
    def update(self, n=1):
        self.n += n
'''

_MPDAF_OBJ = '''\
class Image:
    def __init__(self, *a, **kw):
        raise RuntimeError("mpdaf.obj.Image stand-in: not on the hot path")
'''

_PHOTUTILS = '''\
def detect_sources(*a, **kw):
    raise RuntimeError("photutils stand-in: not on the hot path")
'''


def build_shim(root=None):
    root = root or tempfile.mkdtemp(prefix="origin_ref_shim_")
    mo = os.path.join(root, "muse_origin")
    os.makedirs(mo, exist_ok=True)
    with open(os.path.join(mo, "__init__.py"), "w") as f:
        f.write(_INIT)
    for name in ("lib_origin.py", "source_masks.py"):
        dst = os.path.join(mo, name)
        if not os.path.lexists(dst):
            os.symlink(os.path.join(REFERENCE, "muse_origin", name), dst)
    mp = os.path.join(root, "mpdaf")
    os.makedirs(mp, exist_ok=True)
    for name, text in (("__init__.py", ""), ("tools.py", _MPDAF_TOOLS), ("obj.py", _MPDAF_OBJ)):
        with open(os.path.join(mp, name), "w") as f:
            f.write(text)
    ph = os.path.join(root, "photutils")
    os.makedirs(ph, exist_ok=True)
    with open(os.path.join(ph, "__init__.py"), "w") as f:
        f.write(_PHOTUTILS)
    return root


def load_reference():
    """Return the reference's ``muse_origin.lib_origin`` module."""
    root = build_shim()
    sys.path.insert(0, root)
    os.environ["PYTHONPATH"] = root + os.pathsep + os.environ.get("PYTHONPATH", "")
    import matplotlib
    matplotlib.use("Agg")
    import muse_origin.lib_origin as lib
    return lib


if __name__ == "__main__":
    lib = load_reference()
    print("loaded", lib.__file__)
