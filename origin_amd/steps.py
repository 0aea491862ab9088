"""Step seam (tier B2): the four hot steps of the reference's
``muse_origin/steps.py`` -- ``Preprocessing`` (:355), ``ComputePCAThreshold`` (:572),
``ComputeGreedyPCA`` (:634), ``ComputeTGLR`` (:707) -- with the same ``name``, ``desc``,
``require``, ``DataObj`` labels and ``run`` keyword names/defaults, running on the GPU.

Two ways to use them:

* ``register()``: when ``muse_origin`` is importable, subclasses of the reference's own
  step classes (only ``run`` overridden) are swapped into ``muse_origin.steps.STEPS``
  before ``ORIGIN(...)`` is constructed (origin.py:193 reads that list).
* stand-alone: ``SimpleOrig`` carries exactly the attributes the four ``run`` bodies read
   and the minimal ``Step`` machinery below (status, ``require`` check,
  parameter recording, runtime) mirrors reference steps.py:101-352 so tests and the
  benchmark drive the same code without mpdaf.

Cubes produced by one step stay in HBM for the next one (``LazyCube``); a host float64
copy is only made when somebody reads ``._data``.
"""
import inspect
import logging
import os
import time
from collections import OrderedDict
from datetime import datetime
from enum import Enum

import numpy as np
from scipy import ndimage as ndi
from scipy.signal import fftconvolve

from . import fitsio, kernels, pipeline
from .device import DeviceArray, default_context
from .thresholds import compute_thresh_gaussfit

__all__ = ('Preprocessing', 'CreateAreas', 'ComputePCAThreshold', 'ComputeGreedyPCA', 'ComputeTGLR',
           'Status', 'Step', 'DataObj', 'SimpleOrig', 'STEPS', 'register', 'unregister')


# ----------------------------------------------------------------------------- containers
class LazyCube:
    """What a DataObj holds when mpdaf is absent: a device array plus an on-demand host
    copy with the dtype the reference would have produced.  ``._data`` / ``.data`` mirror
    the mpdaf attribute the ``run`` bodies read (e.g. steps.py:617, :688, :771)."""

    def __init__(self, dev=None, host=None, dtype=np.float64):
        self.dev, self._host, self._dtype = dev, host, dtype

    @property
    def _data(self):
        if self._host is None:
            self._host = (self.dev.to_host_f64() if self._dtype == np.float64
                          else self.dev.to_host().astype(self._dtype, copy=False))
        return self._host

    data = _data

    @property
    def shape(self):
        return self.dev.shape if self.dev is not None else self._host.shape

    def device(self, ctx, dtype=np.float32):
        if self.dev is None:
            self.dev = ctx.to_device(self._host, dtype)
        return self.dev


def _wrap(ctx, value, dtype=np.float32):
    """DeviceArray for anything cube-like a step may be handed (LazyCube, mpdaf object with
    ``_data``, ndarray)."""
    if isinstance(value, (LazyCube, fitsio.FitsCube)):
        return value.device(ctx, dtype)
    if isinstance(value, DeviceArray):
        return value
    if hasattr(value, "dense"):        # sparse.SparseCube
        return value.dense()
    data = getattr(value, "_data", value)
    return ctx.to_device(np.asarray(data), dtype)


# ----------------------------------------------------------------------------- framework
# Stand-alone harness for sessions without muse_origin/mpdaf.  It honours the same contract as
# the reference's Step machinery (steps.py:112-299) -- the persisted keys of ``param[name]``
# ('stepidx', 'params', 'status', 'runtime', 'execution_date'), the ``stepNN_<name>`` method
# names, the ``require`` check and its error text, FAILED on exceptions -- but is built
# differently: descriptors name themselves (``__set_name__``), classes collect their outputs
# along the MRO in ``__init_subclass__`` (so a mixin + base combination keeps them, which the
# reference's metaclass does not: see register()), and reload goes through a reader table.
class Status(Enum):
    """Values are what the reference persists in the session's yaml (steps.py:112-118)."""
    NOTRUN = 'not run yet'
    RUN = 'run'
    DUMPED = 'dumped outputs'
    FAILED = 'failed'


_DONE = (Status.RUN, Status.DUMPED)

_READERS = {
    'cube': lambda path: fitsio.FitsCube(path),
    'image': lambda path: fitsio.FitsCube(path),
    'table': lambda path: fitsio.read_table(path),
    'array': lambda path: np.loadtxt(path, ndmin=1),
}


class DataObj:
    """A named output of a step.  Holds the value, or -- once the step has been dumped -- the
    path of its file, in which case the first read brings it back through ``_READERS`` (cubes
    and images as ``fitsio.FitsCube``, decoded by the GPU when used).  A path whose file is
    gone reads as None, an output never produced too (reference behaviour, steps.py:121-163)."""

    def __init__(self, kind):
        self.kind = kind
        self.label = None

    def __set_name__(self, owner, name):
        self.label = name

    def __get__(self, obj, owner=None):
        if obj is None:
            return None
        slot = vars(obj)
        val = slot.get(self.label)
        if not isinstance(val, str):
            return val
        if not os.path.isfile(val):
            return None
        slot[self.label] = val = _READERS[self.kind](val)
        return val

    def __set__(self, obj, val):
        vars(obj)[self.label] = val


def _outputs_of(cls):
    """(label, kind) of every DataObj visible on ``cls``, base classes first."""
    seen = OrderedDict()
    for klass in reversed(cls.__mro__):
        for label, attr in vars(klass).items():
            if isinstance(attr, DataObj):
                seen[label] = attr.kind
    return list(seen.items())


class Step:
    """One processing step of a session: ``step(**kw)`` records the keyword values, refuses to
    run before the steps named in ``require``, runs, and leaves status / runtime / date in
    ``orig.param[name]``."""

    name = None
    desc = None
    require = None
    _dataobjs = []

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        cls._dataobjs = _outputs_of(cls)

    def __init__(self, orig, idx, param):
        self.logger = logging.getLogger(__name__)
        self.orig, self.idx = orig, idx
        self.method_name = f'step{idx:02d}_{self.name}'
        # a reloaded session brings its own record: keep what is there
        self.meta = param.setdefault(self.name, {})
        self.meta.setdefault('stepidx', idx)
        self.param = self.meta.setdefault('params', {})

    def __repr__(self):
        return f'Step {self.idx:02d}: <{type(self).__name__}(status: {self.status.name})>'

    def _loginfo(self, *args):
        self.logger.info(*args)

    def _logdebug(self, *args):
        self.logger.debug(*args)

    def _logwarning(self, *args):
        self.logger.warning(*args)

    status = property(lambda self: self.meta.get('status', Status.NOTRUN),
                      lambda self, val: self.meta.__setitem__('status', val))

    def _record_keywords(self, kwargs):
        for key, par in inspect.signature(self.run).parameters.items():
            if key != 'orig':
                self.param[key] = kwargs.get(key, par.default)

    def _check_required(self):
        for other in (self.orig.steps[r] for r in self.require or ()):
            if other.status not in _DONE:
                raise RuntimeError(f'step {other.idx:02d} must be run before')

    def __call__(self, *args, **kwargs):
        start = time.time()
        self._loginfo('Step %02d - %s', self.idx, self.desc)
        self._record_keywords(kwargs)
        self._check_required()
        self.status = Status.FAILED          # stays if run() raises
        self.run(self.orig, *args, **kwargs)
        self.status = Status.RUN
        self.meta['runtime'] = time.time() - start
        self.meta['execution_date'] = datetime.now().isoformat()
        self._loginfo('%02d Done - %.2f sec.', self.idx, self.meta['runtime'])

    # without mpdaf the outputs stay LazyCube / ndarray
    def store_cube(self, name, data, **kwargs):
        setattr(self, name, data)

    def store_image(self, name, data, **kwargs):
        setattr(self, name, data)

    def dump(self, outpath):
        """Save the outputs of a step that has been run and replace them by the paths of
        their files (reference steps.py:301-340): ``<outpath>/<name>.fits`` for cubes, images
        and tables, ``<name>.txt`` for arrays.  Cubes go from HBM to the file through the
        device-side FITS encoder (fitsio.write_image), float64 on disk unless the reference
        itself holds another type (``convert_float32=False``, steps.py:319)."""
        if self.status is not Status.RUN:
            return
        ctx = getattr(self.orig, 'hip_ctx', None)
        for name, kind in self._dataobjs:
            obj = getattr(self, name)
            if obj is None:
                continue
            ext = 'txt' if kind == 'array' else 'fits'
            outf = f'{outpath}/{name}.{ext}'
            self.logger.debug('   - %s [%s]', name, kind)
            if kind in ('cube', 'image'):
                fitsio.write_image(outf, obj, ctx=ctx, header=self._wcs_cards(kind))
            elif kind == 'table':
                if hasattr(obj, 'write'):  # astropy Table
                    obj.write(outf, overwrite=True)
                else:
                    fitsio.write_table(outf, OrderedDict(
                        (c, obj[c]) for c in getattr(obj, 'colnames', obj)))
            elif kind == 'array':
                np.savetxt(outf, obj)
            # the attribute becomes the path of its file: the data (and its copy in HBM) is
            # released and comes back from the file when the attribute is read again
            setattr(self, name, outf)
            _cache(self.orig).pop(name, None)
        self.status = Status.DUMPED

    def load(self, outpath):
        """Point the outputs of a dumped step at their files; they are read when accessed
        (reference steps.py:342-352)."""
        if self.status is not Status.DUMPED:
            return
        for name, kind in self._dataobjs:
            ext = 'txt' if kind == 'array' else 'fits'
            setattr(self, name, f'{outpath}/{name}.{ext}')

    def _wcs_cards(self, kind):
        """World-coordinate cards of the session (``orig.wcs_header`` / ``orig.wave_header``:
        plain mappings of FITS keywords when given), as mpdaf adds to the DATA extension."""
        cards = OrderedDict()
        cards.update(getattr(self.orig, 'wcs_header', None) or {})
        if kind == 'cube':
            cards.update(getattr(self.orig, 'wave_header', None) or {})
        return cards


# ----------------------------------------------------------------------------- helpers
def compute_segmap_gauss(data, pfa, fwhm_fsf=0, bins='fd'):
    """2-D host heuristic of reference lib_origin.py:243-280 (outside the hot path; kept so
    that ``Preprocessing`` still fills ``segmap_cont`` / ``segmap_merged``)."""
    histO2, frecO2, gamma, mea, std = compute_thresh_gaussfit(data, pfa, bins=bins)
    mask = data > gamma
    mask = ndi.binary_erosion(mask, border_value=1, iterations=1)
    mask = ndi.binary_dilation(mask, iterations=1)
    if fwhm_fsf > 0:
        fwhm_pix = int(fwhm_fsf) // 2
        size = fwhm_pix * 2 + 1
        disc = np.hypot(*list(np.mgrid[:size, :size] - fwhm_pix)) < fwhm_pix
        mask = fftconvolve(mask, disc, mode='same')
        mask = mask > 1e-9
    return gamma, ndi.label(mask)[0]


def _ctx_of(orig):
    ctx = getattr(orig, 'hip_ctx', None)
    if ctx is None:
        ctx = default_context(0)
        try:
            orig.hip_ctx = ctx
        except Exception:
            pass
    return ctx


def _cache(orig):
    c = orig.__dict__.get('_hip_cache')
    if c is None:
        c = orig.__dict__['_hip_cache'] = {}
    return c


# Devices of the sessions that do not say themselves (``register(devices=[...])`` sets it for
# the reference's ORIGIN objects, which know nothing of GPUs); None = one context on device 0.
_DEFAULT_DEVICES = None


def _session_of(orig):
    """The ``session.TiledSession`` of a session that runs on several devices (``orig.hip_devices``
    -- ``SimpleOrig(..., devices=[...])`` -- or ``register(devices=[...])``), made on first use;
    None for the usual one-context session.  One process, one context and one thread per entry;
    the same ordinal twice means two contexts on that card (strips through the host)."""
    sess = orig.__dict__.get('_hip_session')
    if sess is not None:
        return sess
    devices = orig.__dict__.get('hip_devices', None) or _DEFAULT_DEVICES
    if devices is None or len(devices) < 2:
        return None
    from .session import DeviceGroup, TiledSession
    sess = TiledSession(DeviceGroup(devices, orig.__dict__.get('hip_backend')))
    orig.__dict__['_hip_session'] = sess
    return sess


def _inputs_on_device(orig, ctx):
    """cube_raw / var / mask uploaded once per session (origin.py:262-274)."""
    c = _cache(orig)
    if 'raw' not in c:
        c['raw'] = _wrap(ctx, orig.cube_raw)
        c['var'] = _wrap(ctx, orig.var)
        m = orig.mask
        m = np.zeros(c['raw'].shape, np.uint8) if m is None or m is np.ma.nomask else m
        # (a bool mask goes up as it is: one byte of 0 / 1 per voxel, no host-side copy)
        c['mask'] = _wrap(ctx, np.asarray(m), np.uint8)
    return c['raw'], c['var'], c['mask']


# ----------------------------------------------------------------------------- run bodies
class _HipStepMixin:
    """Store / fetch cubes so that they stay in HBM between steps.  Under the stand-alone
    ``Step`` the DataObj holds a ``LazyCube``; under the reference's ``Step`` (register())
    ``store_cube`` builds an mpdaf Cube from a host array as usual."""

    def _put_cube(self, orig, name, dev, dtype=np.float64):
        _cache(orig)[name] = dev
        if isinstance(self, Step):
            self.store_cube(name, LazyCube(dev, dtype=dtype))
        else:
            self.store_cube(name, dev.to_host_f64() if dtype == np.float64
                            else dev.to_host().astype(dtype, copy=False))

    def _get_cube(self, orig, ctx, name):
        dev = _cache(orig).get(name)
        if dev is not None:
            return dev
        return _wrap(ctx, getattr(orig, name))


class _PreprocessingRun(_HipStepMixin):
    name = 'preprocessing'
    desc = 'Preprocessing'

    def run(self, orig, dct_order=10, dct_approx=False, pfasegcont=0.01, pfasegres=0.01,
            local_max_size=3, bins='fd'):
        sess = _session_of(orig)
        self._loginfo('DCT computation')
        if sess is not None:   # several devices: row bands, one all-reduce, a one-spaxel halo
            out = sess.preprocess(orig.cube_raw, orig.var, orig.mask, dct_order, dct_approx,
                                  local_max_size)
            Nz = out['cube_std'].shape[0]
            lmax, lmin = out['cube_std_local_max'], out['cube_std_local_min']
            ima_std, ima_dct, o2, cont_o2 = (out['ima_std'], out['ima_dct'].astype(np.float32),
                                             out['o2'], out['cont_o2'])
        else:
            ctx = _ctx_of(orig)
            raw, var, mask = _inputs_on_device(orig, ctx)
            Nz = raw.shape[0]
            out = pipeline.preprocess(ctx, raw, var, mask, dct_order, dct_approx,
                                      allreduce=getattr(orig, 'allreduce', None))
            ima_std = out['ima_std'].to_host().astype(np.float64)
            from . import sparse
            lmax, lmin = sparse.local_max(ctx, out['cube_std'], out['cube_std'], mask,
                                          local_max_size)
            ima_dct, o2 = out['ima_dct'].to_host(), out['o2_host']
            # sum_z cont_dct^2 is a per-spaxel reduction of the device cube
            cont_o2 = kernels.o2test(ctx, out['cont_dct']).to_host()
        self._loginfo('Std signal saved in self.cube_std and self.ima_std')
        self._put_cube(orig, 'cube_std', out['cube_std'])
        self.store_image('ima_std', ima_std)

        self._loginfo('Compute local maximum of std cube values')
        self._put_cube(orig, 'cube_std_local_max', lmax)
        self._put_cube(orig, 'cube_std_local_min', lmin)

        self._loginfo('DCT continuum saved in self.cont_dct and self.ima_dct')
        self._put_cube(orig, 'cont_dct', out['cont_dct'], np.float32)
        self.store_image('ima_dct', ima_dct)
        _cache(orig)['o2_std'] = o2

        mean_fwhm = int(np.ceil(np.mean(orig.FWHM_PSF)))
        self._loginfo('Segmentation based on the continuum')
        map1 = np.log10(cont_o2 * Nz)
        thresh, map_cont = compute_segmap_gauss(map1, pfasegcont, mean_fwhm, bins=bins)
        self.store_image('segmap_cont', map_cont)
        self._loginfo('Segmentation based on the residual')
        thresh, map_res = compute_segmap_gauss(o2, pfasegres, mean_fwhm, bins=bins)
        self._loginfo('Merging both maps')
        segmap, nlabels = ndi.label((map_cont > 0) | (map_res > 0))
        self.store_image('segmap_merged', segmap)


class _CreateAreasRun:
    """CreateAreas.run (reference steps.py:492-569): host geometry, origin_amd/areas.py."""
    name = 'areas'
    desc = 'Areas creation'
    _Status = Status     # register() points this at the reference's own enum

    def run(self, orig, pfa=0.2, minsize=100, maxsize=None):
        from .areas import create_areamap
        mask = np.asarray(getattr(orig.mask, '_data', orig.mask), dtype=bool)
        nexp = int(np.sum(mask.shape[0] - mask.sum(axis=0) > 0))
        nsub = np.maximum(1, int(np.sqrt(nexp / (minsize ** 2))))
        if nsub > 1:
            self._loginfo('First segmentation of %d^2 square', nsub)
        areamap, nbAreas = create_areamap(mask, _data(orig.segmap_merged), pfa, minsize, maxsize)
        orig.param['nbareas'] = nbAreas
        self.store_image('areamap', areamap)
        self._loginfo('Save the map of areas in self.areamap')
        self._loginfo('%d areas generated', nbAreas)

    def set_areamap(self, areamap):
        """Use an area map made elsewhere (a regular grid in the benchmarks, a map loaded from a
        previous session) instead of running the step: same outputs and status as ``run``."""
        areamap = np.asarray(getattr(areamap, '_data', areamap)).astype(int)
        labels = np.unique(areamap)
        self.orig.param['nbareas'] = len(labels) - (1 if 0 in labels else 0)
        self.store_image('areamap', areamap)
        self.status = self._Status.RUN


class _ComputePCAThresholdRun(_HipStepMixin):
    name = 'compute_PCA_threshold'
    desc = 'PCA threshold computation'
    require = ('preprocessing', 'areas')

    def run(self, orig, pfa_test=0.01):
        o2 = _cache(orig).get('o2_std')
        if o2 is None:  # session reloaded: recompute the O2 map from cube_std
            ctx = _ctx_of(orig)
            o2 = kernels.o2test(ctx, self._get_cube(orig, ctx, 'cube_std')).to_host()
        areamap = getattr(orig.areamap, '_data', orig.areamap)
        res = pipeline.pca_threshold(o2, areamap, orig.nbAreas, pfa_test)
        for i in range(orig.nbAreas):
            self._loginfo('Area %d, estimation mean/std/threshold: %f/%f/%f', i + 1,
                          res['meaO2'][i], res['stdO2'][i], res['thresO2'][i])
        orig.testO2, orig.histO2, orig.binO2 = res['testO2'], res['histO2'], res['binO2']
        self.thresO2, self.meaO2, self.stdO2 = res['thresO2'], res['meaO2'], res['stdO2']


class _ComputeGreedyPCARun(_HipStepMixin):
    name = 'compute_greedy_PCA'
    desc = 'Greedy PCA computation'
    require = ('preprocessing', 'areas', 'compute_PCA_threshold')

    def run(self, orig, Noise_population=50, itermax=100, threshold_list=None):
        ctx = _ctx_of(orig) if _session_of(orig) is None else None
        thr = orig.thresO2 if threshold_list is None else threshold_list
        orig.param['threshold_list'] = thr
        self._loginfo('   - List of threshold = %s', ' '.join("%.2f" % x for x in thr))
        self._loginfo('Compute greedy PCA on each zone')
        areamap = getattr(orig.areamap, '_data', orig.areamap)
        sess = _session_of(orig)
        if sess is not None:   # several devices: whole areas per rank (session.TiledSession)
            std = _cache(orig).get('cube_std')
            if std is None:
                std = orig.cube_std
            faint, mapO2, nstop = sess.greedy_pca(
                std, areamap, orig.nbAreas, thr, orig.testO2, _glr_halo(orig), Noise_population,
                itermax)
        else:
            faint, mapO2, nstop, drv = pipeline.greedy_pca(
                ctx, self._get_cube(orig, ctx, 'cube_std'), areamap, orig.nbAreas, thr,
                orig.testO2, Noise_population, itermax)
        if nstop > 0:
            self._logwarning('The iterations have been reached the limit of %d in %d cases',
                             itermax, nstop)
        self._put_cube(orig, 'cube_faint', faint)
        self.store_image('mapO2', mapO2)


class _ComputeTGLRRun(_HipStepMixin):
    name = 'compute_TGLR'
    desc = 'GLR test'
    require = ('compute_greedy_PCA',)

    def run(self, orig, size=3, ncpu=1, pcut=1e-8, pmeansub=True):
        sess = _session_of(orig)
        self._loginfo('Correlation')
        if sess is not None:   # several devices: halo exchange + GLR per box of areas
            faint = _cache(orig).get('cube_faint')
            if faint is None:
                faint = orig.cube_faint
            areamap = getattr(orig.areamap, '_data', orig.areamap)
            if getattr(sess, 'host_mask', None) is None:
                sess.host_mask = _host_mask(orig, faint.shape)
            out = sess.tglr(faint, areamap, orig.PSF, orig.wfields, orig.profiles, size, pcut,
                            pmeansub)
            out['maxmap'], out['minmap'] = _HostImage(out['maxmap']), _HostImage(out['minmap'])
        else:
            ctx = _ctx_of(orig)
            _, _, mask = _inputs_on_device(orig, ctx)
            faint = self._get_cube(orig, ctx, 'cube_faint')
            plan = kernels.GLRPlan(ctx, faint.shape, orig.PSF, orig.wfields, orig.profiles, pcut,
                                   pmeansub)
            try:
                out = pipeline.tglr(ctx, plan, faint, mask, size)
                ctx.sync()
            finally:
                plan.close()
        self._put_cube(orig, 'cube_correl', out['correl'])
        self._put_cube(orig, 'cube_correl_min', out['correl_min'])
        self._put_cube(orig, 'cube_profile', out['profile'], np.uint8)
        self.store_image('maxmap', out['maxmap'].to_host().astype(np.float64))
        self.store_image('minmap', out['minmap'].to_host().astype(np.float64))
        self._put_cube(orig, 'cube_local_max', out['local_max'])
        self._put_cube(orig, 'cube_local_min', out['local_min'])


class _ComputePurityThresholdRun(_HipStepMixin):
    """ComputePurityThreshold.run (reference steps.py:848-890): the purity curves come from
    reductions of the local-maximum cubes that are still in HBM."""
    name = 'compute_purity_threshold'
    desc = 'Compute Purity threshold'
    require = ('compute_TGLR',)

    def run(self, orig, purity=0.9, purity_std=None, threshlist=None, pfasegfinal=1e-5,
            bins='fd'):
        from .lib_origin import Compute_threshold_purity
        ctx = _ctx_of(orig)
        if purity_std is None:
            purity_std = purity
        orig.param.update(dict(purity=purity, purity_std=purity_std))
        # another segmap on the maxmap, merged with segmap_merged          (steps.py:862-867)
        thresh, map_res = compute_segmap_gauss(_data(orig.maxmap), pfasegfinal, 0, bins=bins)
        segmap, nlabels = ndi.label((map_res > 0) | (_data(orig.segmap_merged) > 0))
        self.store_image('segmap_purity', segmap)

        self._loginfo('Estimation of threshold with purity = %.2f', purity)
        threshold, pval = Compute_threshold_purity(
            purity, self._get_cube(orig, ctx, 'cube_local_max'),
            self._get_cube(orig, ctx, 'cube_local_min'), segmap, threshlist=threshlist)
        self.Pval = pval
        orig.param['threshold'] = threshold
        self._loginfo('Threshold: %.2f ', threshold)

        self._loginfo('Estimation of threshold std with purity = %.2f', purity_std)
        threshold_std, pval = Compute_threshold_purity(
            purity_std, self._get_cube(orig, ctx, 'cube_std_local_max'),
            self._get_cube(orig, ctx, 'cube_std_local_min'), threshlist=threshlist)
        self.Pval_comp = pval
        orig.param['threshold_std'] = threshold_std
        self._loginfo('Threshold: %.2f ', threshold_std)


def _data(img):
    """ndarray of an image attribute (mpdaf Image under the reference, ndarray here)."""
    return np.asarray(getattr(img, '_data', img))


class _HostImage:
    """A map that is already on the host, where the run bodies expect ``.to_host()``."""

    def __init__(self, arr):
        self._a = np.asarray(arr)

    def to_host(self):
        return self._a


def _host_mask(orig, shape):
    m = orig.mask
    if m is None or m is np.ma.nomask:
        return np.zeros(shape, np.uint8)
    return np.asarray(getattr(m, '_data', m))


def _glr_halo(orig, size=3):
    """Spaxels the GLR of step 5 and its local maxima read beyond a rank's own: half the PSF's
    side (PSF_size, origin.py:161) plus half the maximum filter's (steps.py:756 ``size=3``)."""
    psf = orig.PSF[0] if isinstance(orig.PSF, (list, tuple)) else orig.PSF
    return int(np.asarray(psf).shape[-1]) // 2 + int(size) // 2


# ----------------------------------------------------------------------------- stand-alone
class Preprocessing(_PreprocessingRun, Step):
    cube_std = DataObj('cube')
    cont_dct = DataObj('cube')
    ima_std = DataObj('image')
    ima_dct = DataObj('image')
    segmap_cont = DataObj('image')
    segmap_merged = DataObj('image')
    cube_std_local_min = DataObj('cube')
    cube_std_local_max = DataObj('cube')


class CreateAreas(_CreateAreasRun, Step):
    areamap = DataObj('image')



class ComputePCAThreshold(_ComputePCAThresholdRun, Step):
    thresO2 = DataObj('array')
    meaO2 = DataObj('array')
    stdO2 = DataObj('array')


class ComputeGreedyPCA(_ComputeGreedyPCARun, Step):
    cube_faint = DataObj('cube')
    mapO2 = DataObj('image')


class ComputeTGLR(_ComputeTGLRRun, Step):
    cube_correl = DataObj('cube')
    cube_correl_min = DataObj('cube')
    cube_profile = DataObj('cube')
    cube_local_min = DataObj('cube')
    cube_local_max = DataObj('cube')
    maxmap = DataObj('image')
    minmap = DataObj('image')


class ComputePurityThreshold(_ComputePurityThresholdRun, Step):
    __doc__ = _ComputePurityThresholdRun.__doc__
    Pval = DataObj('table')
    Pval_comp = DataObj('table')
    segmap_purity = DataObj('image')


STEPS = [Preprocessing, CreateAreas, ComputePCAThreshold, ComputeGreedyPCA, ComputeTGLR,
         ComputePurityThreshold]


class SimpleOrig:
    """Stand-in for the ``ORIGIN`` session object carrying exactly what the four hot ``run``
    bodies read: cube_raw, var, mask, FWHM_PSF, PSF, wfields, profiles,
    nbAreas, areamap, param, testO2, thresO2, cube_std, cube_faint.  Steps are bound as
    ``stepNN_name`` callables like origin.py:193-208 does."""

    def __init__(self, cube_raw, var, mask, PSF, profiles, FWHM_PSF=3.3, wfields=None,
                 param=None, ctx=None, devices=None, backend=None):
        """``devices``: None (one context, ``ctx`` or device 0) or a list of device ordinals --
        the hot steps then run tiled over them in this one process (origin_amd/session.py); the
        same ordinal twice = two contexts on one card.  ``backend``: "rccl" / "host" for the
        cubes that travel between them (default: RCCL when the ordinals differ)."""
        self.hip_devices = list(devices) if devices is not None else None
        self.hip_backend = backend
        self.cube_raw, self.var, self.mask = cube_raw, var, mask
        self.PSF, self.wfields, self.profiles = PSF, wfields, profiles
        self.FWHM_PSF = FWHM_PSF
        self.param = param or {}
        self.wave = self.wcs = None
        self.testO2 = self.histO2 = self.binO2 = None
        self.hip_ctx = ctx or default_context(0 if not devices else int(devices[0]))
        self.steps = OrderedDict()
        self._dataobjs = {}
        for i, cls in enumerate(STEPS, start=1):
            step = cls(self, i, self.param)
            self.steps[step.name] = step
            self.__dict__[step.method_name] = step
            for name, _ in step._dataobjs:
                self._dataobjs[name] = step

    def __getattr__(self, name):
        d = self.__dict__.get('_dataobjs', {})
        if name in d:
            return getattr(d[name], name)
        raise AttributeError(f"unknown attribute {name}")

    @property
    def nbAreas(self):
        return self.param.get('nbareas')


# ----------------------------------------------------------------------------- register
def register(devices=None):
    """Swap GPU versions of the six steps built here into ``muse_origin.steps.STEPS``.  Call
    before constructing ``ORIGIN`` (origin.py:193 reads the list then).  Returns the list of
    replaced class names.  ``devices``: list of device ordinals the sessions made afterwards
    spread their hot steps over (one process, one context per entry: origin_amd/session.py);
    None = one context on device 0.

    Each replacement is ``class <Name>(<run mixin>, <reference class>)`` made with the
    reference's own metaclass.  That metaclass rebuilds ``_dataobjs`` from the attributes of
    the class body alone (steps.py:176-185), which is empty here -- the DataObj descriptors
    are inherited -- so the list is copied over from the reference class afterwards;
    ``ORIGIN.__init__`` (origin.py:206-207), ``Step.dump`` and ``Step.load`` all walk it."""
    import muse_origin.steps as ref  # noqa: raises ImportError when the reference is absent

    global _DEFAULT_DEVICES
    _DEFAULT_DEVICES = list(devices) if devices is not None else None
    replaced = []
    for mixin, refname in ((_PreprocessingRun, 'Preprocessing'),
                           (_CreateAreasRun, 'CreateAreas'),
                           (_ComputePCAThresholdRun, 'ComputePCAThreshold'),
                           (_ComputeGreedyPCARun, 'ComputeGreedyPCA'),
                           (_ComputeTGLRRun, 'ComputeTGLR'),
                           (_ComputePurityThresholdRun, 'ComputePurityThreshold')):
        base = getattr(ref, refname)
        if getattr(base, '_origin_amd_base', None) is not None:   # already registered
            replaced.append(refname)
            continue
        new = type(base)(refname, (mixin, base), {'__doc__': base.__doc__,
                                                  '__module__': base.__module__})
        new._dataobjs = list(base._dataobjs)
        new._Status = ref.Status
        new._origin_amd_base = base
        ref.STEPS[ref.STEPS.index(base)] = new
        setattr(ref, refname, new)
        replaced.append(refname)
    return replaced


def unregister():
    """Put the reference's own classes back (tests; A/B runs of one session)."""
    import muse_origin.steps as ref
    for i, cls in enumerate(list(ref.STEPS)):
        base = getattr(cls, '_origin_amd_base', None)
        if base is not None:
            ref.STEPS[i] = base
            setattr(ref, base.__name__, base)
