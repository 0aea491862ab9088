"""TEST INFRASTRUCTURE ONLY -- never imported by the product, never shipped to the GPU box.

Pins ``origin_amd.steps.register()`` on the reference's REAL ``muse_origin/steps.py`` (B2 seam,
SURVEY.md 8b).  Build container only: /opt/conda/bin/python3.9 (numpy / scipy /
astropy; no GPU needed -- nothing here computes).

    /opt/conda/bin/python3.9 oracle/check_register.py [--report oracle/REGISTER_REPORT.txt]

``import muse_origin`` fails in this checkout for ordinary reasons (no generated version.py, no
mpdaf, no photutils).  The shim of oracle/ref_import.py is extended by a symlink to the
reference's steps.py and inert ``mpdaf.obj.Cube / Image / Spectrum`` stand-ins -- names steps.py
imports at module level (steps.py:15) and only USES inside store_cube / store_image / dump /
load.  Nothing of the reference is copied; its Step / StepMeta / DataObj / STEPS run unmodified.

What is checked, against the reference's own objects:
  1. the module imports; StepMeta, Step, DataObj, Status, 11 STEPS (steps.py:1336-1348);
  2. register() swaps exactly six classes; each replacement is a subclass of the reference's
     class made by the reference's metaclass, keeps name / desc / require / _dataobjs and the
     keyword names AND defaults of ``run`` (introspected and persisted, steps.py:255-263);
  3. all 11 steps instantiate the way ORIGIN.__init__ does (origin.py:193-208): method name
     ``stepNN_<name>``, DataObj labels visible as attributes (None before the step has run);
  4. Step.__call__ (the reference's, unmodified) drives a replacement whose ``run`` is patched
     to a recorder: parameters land in ``param``, status goes NOTRUN -> RUN, ``require`` is
     enforced against a step that has not run, an exception in ``run`` gives FAILED and is
     re-raised (steps.py:242-281);
  5. unregister() puts the reference's classes back, in place.
"""
import argparse
import inspect
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402

_MPDAF_OBJ = '''\
class _Inert:
    """Stand-in for an mpdaf data class: keeps what it was given (store_cube / store_image)."""
    def __init__(self, *a, **kw):
        self.args, self.kw = a, kw
        self._data = kw.get("data")
class Cube(_Inert): pass
class Image(_Inert): pass
class Spectrum(_Inert): pass
'''

SIX = ('Preprocessing', 'CreateAreas', 'ComputePCAThreshold', 'ComputeGreedyPCA', 'ComputeTGLR',
       'ComputePurityThreshold')


def load_reference_steps():
    root = ref_import.build_shim(tempfile.mkdtemp(prefix="origin_ref_steps_"))
    with open(os.path.join(root, "mpdaf", "obj.py"), "w") as f:
        f.write(_MPDAF_OBJ)
    dst = os.path.join(root, "muse_origin", "steps.py")
    if not os.path.lexists(dst):
        os.symlink(os.path.join(ref_import.REFERENCE, "muse_origin", "steps.py"), dst)
    sys.path.insert(0, root)
    import matplotlib
    matplotlib.use("Agg")
    import muse_origin.steps as ref
    assert os.path.realpath(ref.__file__) == os.path.realpath(
        os.path.join(ref_import.REFERENCE, "muse_origin", "steps.py"))
    return ref


class _Session:
    """The part of ORIGIN.__init__ that builds the steps (origin.py:193-208), verbatim in
    behaviour: one instance per class of STEPS, bound as stepNN_<name>, DataObj labels
    forwarded to the step that owns them (origin.py:246-253)."""

    def __init__(self, ref):
        self.param = {}
        self.steps = {}
        self._dataobjs = {}
        self.wave = self.wcs = None
        for i, cls in enumerate(ref.STEPS, start=1):
            step = cls(self, i, self.param)
            self.steps[step.name] = step
            setattr(self, step.method_name, step)
            for name, _ in step._dataobjs:
                self._dataobjs[name] = step

    def __getattr__(self, name):
        d = self.__dict__.get("_dataobjs", {})
        if name in d:
            return getattr(d[name], name)
        raise AttributeError(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--report", default=None)
    args = ap.parse_args()
    out = []

    def say(msg):
        out.append(msg)
        print(msg)

    ref = load_reference_steps()
    say(f"reference module: {os.path.realpath(ref.__file__)}")
    assert isinstance(ref.Step, ref.StepMeta) and len(ref.STEPS) == 11
    before = {n: getattr(ref, n) for n in SIX}
    names_before = [c.name for c in ref.STEPS]
    say(f"STEPS: {len(ref.STEPS)} classes, metaclass {type(ref.Step).__name__}")

    from origin_amd import steps as hip_steps
    replaced = hip_steps.register()
    assert replaced == list(SIX), replaced
    assert [c.name for c in ref.STEPS] == names_before
    for n in SIX:
        new, old = getattr(ref, n), before[n]
        assert new is not old and issubclass(new, old) and type(new) is ref.StepMeta
        assert ref.STEPS[names_before.index(old.name)] is new
        assert (new.__name__, new.name, new.desc, new.require) == \
            (n, old.name, old.desc, old.require)
        assert new._dataobjs == old._dataobjs and new._dataobjs
        ours = inspect.signature(new.run).parameters
        theirs = inspect.signature(old.run).parameters
        assert [(k, p.default) for k, p in ours.items()] == \
            [(k, p.default) for k, p in theirs.items()], n
        say(f"  {n:24s} name={new.name!r:26s} require={new.require!r} "
            f"dataobjs={[d for d, _ in new._dataobjs]} run{tuple(k for k in ours if k != 'self')}")
    assert hip_steps.register() == list(SIX)          # idempotent
    say("register(): six classes swapped in place, idempotent; the other five untouched: "
        f"{[c.__name__ for c in ref.STEPS if c.__name__ not in SIX]}")

    # ---- instantiate all 11 as ORIGIN.__init__ does
    orig = _Session(ref)
    assert len(orig.steps) == 11
    for i, cls in enumerate(ref.STEPS, start=1):
        st = orig.steps[cls.name]
        assert st.method_name == 'step%02d_%s' % (i, cls.name) and st.status is ref.Status.NOTRUN
        for label, _ in cls._dataobjs:
            assert getattr(orig, label) is None, label   # known, not produced yet
    say(f"session: 11 steps instantiated ({', '.join(s.method_name for s in orig.steps.values())})")

    # ---- the reference's Step.__call__ drives a replacement
    calls = []
    pre_cls = getattr(ref, 'Preprocessing')
    thr_cls = getattr(ref, 'ComputePCAThreshold')
    assert pre_cls.__call__ is ref.Step.__call__        # not overridden: the reference's own

    def rec(self, orig_, **kw):
        calls.append((type(self).__name__, kw))

    def boom(self, orig_, pfa_test=0.01):
        raise ValueError("run failed")

    saved = pre_cls.run, thr_cls.run
    try:
        # require: step 3 needs 'preprocessing' and 'areas' (steps.py:608)
        try:
            orig.step03_compute_PCA_threshold()
            raise AssertionError("require was not enforced")
        except RuntimeError as exc:
            say(f"require enforced by the reference's __call__: {exc}")
        sig = inspect.signature(saved[0])
        pre_cls.run = rec
        rec.__signature__ = sig
        orig.step01_preprocessing(dct_order=7)
        st = orig.steps['preprocessing']
        assert calls == [('Preprocessing', {'dct_order': 7})]
        assert st.status is ref.Status.RUN and st.param['dct_order'] == 7
        assert st.param['dct_approx'] == sig.parameters['dct_approx'].default
        assert 'runtime' in st.meta and 'execution_date' in st.meta
        say(f"__call__ -> run: status RUN, params recorded {dict(st.param)}")
        orig.steps['areas'].status = ref.Status.RUN
        thr_cls.run = boom
        try:
            orig.step03_compute_PCA_threshold()
            raise AssertionError("exception swallowed")
        except ValueError:
            assert orig.steps['compute_PCA_threshold'].status is ref.Status.FAILED
        say("exception in run: status FAILED, re-raised")
    finally:
        pre_cls.run, thr_cls.run = saved

    # register(devices=[...]): the sessions made afterwards spread their hot steps over those
    # devices (origin_amd/session.py); the swap itself is the same, and register() resets it
    assert hip_steps.register(devices=[0, 1]) == list(SIX) and hip_steps._DEFAULT_DEVICES == [0, 1]

    class _O:        # (what _session_of reads of a reference session: its instance dict)
        pass
    o_ = _O()
    assert hip_steps.register() == list(SIX) and hip_steps._DEFAULT_DEVICES is None
    assert hip_steps._session_of(o_) is None
    say("register(devices=[0, 1]) keeps the swap and sets the devices of later sessions; "
        "register() resets them")

    hip_steps.unregister()
    assert [getattr(ref, n) for n in SIX] == [before[n] for n in SIX]
    assert [c for c in ref.STEPS if c.__name__ in SIX] == [before[n] for n in SIX]
    say("unregister(): the reference's classes are back in STEPS and in the module")
    say("OK")
    if args.report:
        with open(args.report, "w") as f:
            f.write("oracle/check_register.py -- register() against the reference's real steps.py\n")
            f.write("(regenerate: /opt/conda/bin/python3.9 oracle/check_register.py --report "
                    "oracle/REGISTER_REPORT.txt)\n\n" + "\n".join(out) + "\n")


if __name__ == "__main__":
    main()
