"""Step seam B2 through ``register()``: the GPU steps swapped into a reference-shaped
``muse_origin.steps`` (tests/_refstub.py reproduces the behaviour of the reference's metaclass,
Step base, STEPS list and ORIGIN attribute passthrough -- reference steps.py:166-299,
:1336-1348, origin.py:193-208,:246-253 -- which cannot be imported without mpdaf)."""
import inspect

import numpy as np
import pytest

from origin_amd import steps as hip_steps
from origin_amd import synth

import _refstub

SIX = ('Preprocessing', 'CreateAreas', 'ComputePCAThreshold', 'ComputeGreedyPCA', 'ComputeTGLR',
       'ComputePurityThreshold')


@pytest.fixture
def ref():
    mod = _refstub.install()
    yield mod
    _refstub.uninstall()


def test_register_swaps_six_classes_and_keeps_their_outputs(ref):
    before = {n: getattr(ref, n) for n in SIX}
    assert hip_steps.register() == list(SIX)
    assert len(ref.STEPS) == 11
    for i, n in enumerate(SIX):
        new = getattr(ref, n)
        assert ref.STEPS[i] is new and new is not before[n]
        assert issubclass(new, before[n]) and type(new) is ref.StepMeta
        assert new.__name__ == n and new.name == before[n].name
        assert new.require == before[n].require
        # the reference's metaclass would leave this empty for a class body without DataObj
        assert new._dataobjs == before[n]._dataobjs and new._dataobjs
        # keyword names and defaults are introspected and persisted (steps.py:255-263)
        ours = inspect.signature(new.run).parameters
        theirs = inspect.signature(before[n].run).parameters
        assert [(k, p.default) for k, p in ours.items()] == \
            [(k, p.default) for k, p in theirs.items()]
    # idempotent, and reversible
    assert hip_steps.register() == list(SIX)
    assert [getattr(ref, n)._origin_amd_base for n in SIX] == [before[n] for n in SIX]
    hip_steps.unregister()
    assert [getattr(ref, n) for n in SIX] == [before[n] for n in SIX]
    assert ref.STEPS[:6] == [before[n] for n in SIX]


def test_session_built_after_register_exposes_the_outputs(ref):
    hip_steps.register()
    f, raw, var, mask = synth.small_case(Nz=80, Ny=12, Nx=12, seed=1, psf_size=5, nprof=2,
                                         area_size=6)
    orig = _refstub.Session(ref, raw, var, mask, f.PSF, f.profiles)
    for label in ('cube_std', 'cont_dct', 'segmap_merged', 'areamap', 'thresO2', 'cube_faint',
                  'mapO2', 'cube_correl', 'maxmap', 'Pval'):
        assert getattr(orig, label) is None           # known, not produced yet
    with pytest.raises(AttributeError):
        orig.no_such_output
    assert list(orig.steps)[:6] == ['preprocessing', 'areas', 'compute_PCA_threshold',
                                    'compute_greedy_PCA', 'compute_TGLR',
                                    'compute_purity_threshold']
    # require is enforced by the reference's own __call__ before our run() is reached
    with pytest.raises(RuntimeError, match="step 01 must be run before"):
        orig.step03_compute_PCA_threshold()
    assert orig.steps['compute_PCA_threshold'].status is ref.Status.NOTRUN
    # set_areamap marks the step with the REFERENCE's enum (require compares against it)
    orig.step02_areas.set_areamap(f.areamap)
    assert orig.steps['areas'].status is ref.Status.RUN
    assert isinstance(orig.areamap, _refstub.Image) and orig.nbAreas == f.nbAreas


def test_standalone_steps_collect_outputs_without_a_metaclass():
    assert type(hip_steps.Step) is type
    assert dict(hip_steps.ComputeTGLR._dataobjs)['cube_profile'] == 'cube'
    assert [n for n, _ in hip_steps.Preprocessing._dataobjs][:2] == ['cube_std', 'cont_dct']

    class Twice(hip_steps.ComputeGreedyPCA):          # inherited outputs stay listed
        pass
    assert Twice._dataobjs == hip_steps.ComputeGreedyPCA._dataobjs


@pytest.mark.gpu
def test_registered_chain_equals_standalone_chain(ref):
    """The whole chain driven through the reference-shaped session (its ``Step.__call__``,
    its ``store_cube`` wrappers, ``orig.__getattr__`` passthrough) gives the arrays the
    stand-alone chain gives, bit for bit."""
    hip_steps.register()
    f, raw, var, mask = synth.small_case(Nz=160, Ny=48, Nx=52, seed=3, psf_size=9, nprof=3,
                                         area_size=24)
    psf = f.PSF.astype(float)
    a = _refstub.Session(ref, raw, var, mask, psf, f.profiles)
    b = hip_steps.SimpleOrig(raw, var, mask, psf, f.profiles)
    for o in (a, b):
        o.step01_preprocessing()
        o.step02_areas.set_areamap(f.areamap)
        o.step03_compute_PCA_threshold()
        o.step04_compute_greedy_PCA()
        o.step05_compute_TGLR()
        o.step06_compute_purity_threshold()
    assert all(s.status is ref.Status.RUN for s in list(a.steps.values())[:6])
    assert isinstance(a.cube_std, _refstub.Cube) and a.cube_std.wave == "wave-coord"
    assert a.cube_faint._data.dtype == np.float64 and a.cont_dct._data.dtype == np.float32
    assert a.cube_profile._data.dtype == np.uint8
    for label in ('cube_std', 'cont_dct', 'cube_faint', 'cube_correl', 'cube_correl_min',
                  'cube_profile', 'cube_local_max', 'cube_local_min', 'cube_std_local_max'):
        np.testing.assert_array_equal(getattr(a, label)._data, getattr(b, label)._data, label)
    for label in ('ima_std', 'ima_dct', 'segmap_merged', 'mapO2', 'maxmap', 'minmap',
                  'segmap_purity'):
        np.testing.assert_array_equal(np.asarray(getattr(a, label)._data),
                                      np.asarray(getattr(b, label)), label)
    np.testing.assert_array_equal(a.thresO2, b.thresO2)
    assert a.param['threshold'] == b.param['threshold']
    assert a.param['compute_TGLR']['params'] == dict(size=3, ncpu=1, pcut=1e-8, pmeansub=True)


@pytest.mark.gpu
def test_registered_chain_on_two_contexts(ref):
    """``register(devices=[0, 0])``: the reference-shaped session (its own ``Step.__call__`` and
    ``store_cube``) drives the hot steps tiled over two contexts of the card -- the one changed
    line of INTEGRATION section 2 -- and ends with the cubes of the one-device chain (to the
    rounding of the all-reduced mean and of another tile geometry in the GLR)."""
    f, raw, var, mask = synth.small_case(Nz=160, Ny=48, Nx=52, seed=3, psf_size=9, nprof=3,
                                         area_size=24)
    psf = f.PSF.astype(float)
    try:
        hip_steps.register(devices=[0, 0])
        a = _refstub.Session(ref, raw, var, mask, psf, f.profiles)
        for step in ("step01_preprocessing",):
            getattr(a, step)()
        a.step02_areas.set_areamap(f.areamap)
        a.step03_compute_PCA_threshold()
        a.step04_compute_greedy_PCA()
        a.step05_compute_TGLR()
        a.step06_compute_purity_threshold()
        assert a.__dict__["_hip_session"].world == 2
    finally:
        hip_steps.register()          # (sessions made from here on: one context again)
    b = hip_steps.SimpleOrig(raw, var, mask, psf, f.profiles)
    b.step01_preprocessing()
    b.step02_areas.set_areamap(f.areamap)
    b.step03_compute_PCA_threshold()
    b.step04_compute_greedy_PCA()
    b.step05_compute_TGLR()
    b.step06_compute_purity_threshold()
    assert all(s.status is ref.Status.RUN for s in list(a.steps.values())[:6])
    assert a.cube_faint._data.dtype == np.float64 and a.cube_profile._data.dtype == np.uint8
    np.testing.assert_array_equal(np.asarray(a.mapO2._data), np.asarray(b.mapO2))
    for label, tol in (('cube_std', 1e-6), ('cube_faint', 1e-6), ('cube_correl', 1e-4),
                       ('cube_correl_min', 1e-4)):
        assert np.max(np.abs(getattr(a, label)._data - getattr(b, label)._data)) <= tol, label
    assert np.max(np.abs(np.asarray(a.maxmap._data) - np.asarray(b.maxmap))) <= 1e-4
    assert np.isclose(a.param['threshold'], b.param['threshold'], rtol=1e-5, equal_nan=True)
    a.__dict__["_hip_session"].close()


def test_register_against_the_real_reference_module():
    """oracle/check_register.py: register() on the reference's REAL muse_origin/steps.py
    (metaclass, Step.__call__, STEPS, the way ORIGIN.__init__ instantiates the steps) -- build
    container only: needs /root/reference and the conda interpreter that has astropy; the
    reference never travels to the GPU box, where this test skips itself."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    py = "/opt/conda/bin/python3.9"
    if not (os.path.exists("/root/reference/muse_origin/steps.py") and os.path.exists(py)):
        pytest.skip("the reference / its interpreter are not on this machine")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([py, os.path.join(root, "oracle", "check_register.py")], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert out.strip().endswith("OK")
    assert "reference module: /root/reference/muse_origin/steps.py" in out
