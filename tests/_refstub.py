"""A stand-in for the parts of ``muse_origin`` that ``origin_amd.steps.register()`` meets, for
boxes where the reference and mpdaf are absent (this container's default interpreter and the
GPU box).  Written for these tests, not taken from the reference: it only reproduces the
BEHAVIOUR that register() must survive --

* a metaclass that labels ``DataObj`` descriptors and lists ONLY the ones in the class body
  under ``_dataobjs`` (reference steps.py:166-185: inherited descriptors are not listed),
* ``Step.store_cube / store_image`` wrapping host arrays in objects with ``._data``, wave/wcs
  taken from the session (steps.py:284-299),
* a mutable module-level ``STEPS`` with eleven entries (steps.py:1336-1348),
* a session object that instantiates ``STEPS`` at construction, binds ``stepNN_<name>``
  callables, exposes step outputs through ``__getattr__`` and fails on unknown names
  (origin.py:193-208, :246-253), keeps ``param['nbareas']`` behind ``nbAreas`` (:491).

``install()`` puts it in ``sys.modules`` as ``muse_origin`` / ``muse_origin.steps``;
``uninstall()`` removes it.
"""
import inspect
import logging
import sys
import time
import types
from collections import OrderedDict
from datetime import datetime
from enum import Enum

import numpy as np


class _Wrapped:
    """What mpdaf's Cube / Image are to the steps: ``._data`` (ndarray) plus coordinates."""

    def __init__(self, data=None, wave=None, wcs=None, mask=None, copy=True, **kw):
        self._data = np.array(data) if copy else np.asarray(data)
        self.wave, self.wcs, self.extra = wave, wcs, kw

    @property
    def data(self):
        return self._data

    @property
    def shape(self):
        return self._data.shape


class Cube(_Wrapped):
    pass


class Image(_Wrapped):
    pass


def _make_steps_module():
    m = types.ModuleType("muse_origin.steps")

    class Status(Enum):
        NOTRUN = 'not run yet'
        RUN = 'run'
        DUMPED = 'dumped outputs'
        FAILED = 'failed'

    class DataObj:
        def __init__(self, kind):
            self.kind = kind

        def __get__(self, obj, owner=None):
            return None if obj is None else obj.__dict__.get(self.label)

        def __set__(self, obj, val):
            obj.__dict__[self.label] = val

    class StepMeta(type):
        # own class body only: a subclass that adds no DataObj gets an EMPTY list
        def __new__(mcs, name, bases, ns):
            own = []
            for key, val in ns.items():
                if isinstance(val, DataObj):
                    val.label = key
                    own.append((key, val.kind))
            ns['_dataobjs'] = own
            return super().__new__(mcs, name, bases, ns)

    class Step(metaclass=StepMeta):
        name = desc = require = None

        def __init__(self, orig, idx, param):
            self.logger = logging.getLogger("muse_origin.stub")
            self.orig, self.idx = orig, idx
            self.method_name = 'step%02d_%s' % (idx, self.name)
            self.meta = param.setdefault(self.name, {})
            self.meta.setdefault('stepidx', idx)
            self.param = self.meta.setdefault('params', {})

        def _loginfo(self, *a):
            self.logger.info(*a)

        _logdebug = _logwarning = _loginfo

        @property
        def status(self):
            return self.meta.get('status', Status.NOTRUN)

        @status.setter
        def status(self, v):
            self.meta['status'] = v

        def __call__(self, *args, **kwargs):
            t0 = time.time()
            for k, p in inspect.signature(self.run).parameters.items():
                if k != 'orig':
                    self.param[k] = kwargs.get(k, p.default)
            for req in self.require or ():
                s = self.orig.steps[req]
                if s.status not in (Status.RUN, Status.DUMPED):
                    raise RuntimeError(f'step {s.idx:02d} must be run before')
            try:
                self.run(self.orig, *args, **kwargs)
            except Exception:
                self.status = Status.FAILED
                raise
            self.status = Status.RUN
            self.meta['runtime'] = time.time() - t0
            self.meta['execution_date'] = datetime.now().isoformat()

        def store_cube(self, name, data, **kw):
            setattr(self, name, Cube(data=data, wave=self.orig.wave, wcs=self.orig.wcs,
                                     mask=np.ma.nomask, copy=False, **kw))

        def store_image(self, name, data, **kw):
            setattr(self, name, Image(data=data, wcs=self.orig.wcs, copy=False, **kw))

        def run(self, orig, **kw):
            raise NotImplementedError("stub of the reference's CPU step")

    def step(clsname, name, outputs, require=None, kwargs=()):
        """A reference-shaped step class: contract attributes only."""
        ns = {'name': name, 'desc': clsname, 'require': require}
        ns.update({label: DataObj(kind) for label, kind in outputs})
        params = [inspect.Parameter('self', inspect.Parameter.POSITIONAL_OR_KEYWORD),
                  inspect.Parameter('orig', inspect.Parameter.POSITIONAL_OR_KEYWORD)]
        params += [inspect.Parameter(k, inspect.Parameter.POSITIONAL_OR_KEYWORD, default=d)
                   for k, d in kwargs]

        def run(self, orig, **kw):
            raise NotImplementedError("stub of the reference's CPU step")
        run.__signature__ = inspect.Signature(params)
        ns['run'] = run
        return StepMeta(clsname, (Step,), ns)

    cube, image, array, table = 'cube', 'image', 'array', 'table'
    m.Preprocessing = step('Preprocessing', 'preprocessing', [
        ('cube_std', cube), ('cont_dct', cube), ('ima_std', image), ('ima_dct', image),
        ('segmap_cont', image), ('segmap_merged', image), ('cube_std_local_min', cube),
        ('cube_std_local_max', cube)],
        kwargs=[('dct_order', 10), ('dct_approx', False), ('pfasegcont', 0.01),
                ('pfasegres', 0.01), ('local_max_size', 3), ('bins', 'fd')])
    m.CreateAreas = step('CreateAreas', 'areas', [('areamap', image)],
                         kwargs=[('pfa', 0.2), ('minsize', 100), ('maxsize', None)])
    m.ComputePCAThreshold = step('ComputePCAThreshold', 'compute_PCA_threshold', [
        ('thresO2', array), ('meaO2', array), ('stdO2', array)],
        require=('preprocessing', 'areas'), kwargs=[('pfa_test', 0.01)])
    m.ComputeGreedyPCA = step('ComputeGreedyPCA', 'compute_greedy_PCA', [
        ('cube_faint', cube), ('mapO2', image)],
        require=('preprocessing', 'areas', 'compute_PCA_threshold'),
        kwargs=[('Noise_population', 50), ('itermax', 100), ('threshold_list', None)])
    m.ComputeTGLR = step('ComputeTGLR', 'compute_TGLR', [
        ('cube_correl', cube), ('cube_correl_min', cube), ('cube_profile', cube),
        ('cube_local_min', cube), ('cube_local_max', cube), ('maxmap', image), ('minmap', image)],
        require=('compute_greedy_PCA',),
        kwargs=[('size', 3), ('ncpu', 1), ('pcut', 1e-8), ('pmeansub', True)])
    m.ComputePurityThreshold = step('ComputePurityThreshold', 'compute_purity_threshold', [
        ('Pval', table), ('Pval_comp', table), ('segmap_purity', image)],
        require=('compute_TGLR',),
        kwargs=[('purity', 0.9), ('purity_std', None), ('threshlist', None),
                ('pfasegfinal', 1e-5), ('bins', 'fd')])
    rest = [step(n, n.lower(), [(n.lower() + '_out', table)])
            for n in ('Detection', 'ComputeSpectra', 'CleanResults', 'CreateMasks', 'SaveSources')]
    m.STEPS = [m.Preprocessing, m.CreateAreas, m.ComputePCAThreshold, m.ComputeGreedyPCA,
               m.ComputeTGLR, m.ComputePurityThreshold] + rest
    m.Status, m.DataObj, m.StepMeta, m.Step = Status, DataObj, StepMeta, Step
    return m


class Session:
    """The session object the steps see, shaped like the reference's ORIGIN for them."""

    def __init__(self, steps_module, cube_raw, var, mask, PSF, profiles, FWHM_PSF=3.3,
                 wfields=None, param=None):
        self.param = param or {}
        self.steps = OrderedDict()
        self._dataobjs = {}
        for i, cls in enumerate(steps_module.STEPS, start=1):
            st = cls(self, i, self.param)
            self.steps[st.name] = st
            self.__dict__[st.method_name] = st
            for label, _ in st._dataobjs:
                self._dataobjs[label] = st
        self.cube_raw, self.var, self.mask = cube_raw, var, mask
        self.Nz, self.Ny, self.Nx = self.shape = cube_raw.shape
        self.wave, self.wcs = "wave-coord", "wcs-coord"
        self.PSF, self.wfields, self.profiles, self.FWHM_PSF = PSF, wfields, profiles, FWHM_PSF
        self.testO2 = self.histO2 = self.binO2 = None

    def __getattr__(self, name):
        if name in self._dataobjs:
            return getattr(self._dataobjs[name], name)
        raise AttributeError(f"unknown attribute {name}")

    @property
    def nbAreas(self):
        return self.param.get("nbareas")


def install():
    pkg = types.ModuleType("muse_origin")
    pkg.__path__ = []
    pkg.steps = _make_steps_module()
    sys.modules["muse_origin"] = pkg
    sys.modules["muse_origin.steps"] = pkg.steps
    return pkg.steps


def uninstall():
    sys.modules.pop("muse_origin.steps", None)
    sys.modules.pop("muse_origin", None)
