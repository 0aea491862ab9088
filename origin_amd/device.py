"""Device context and arrays on top of the C ABI (no PyTorch, no numpy-on-GPU library)."""
import atexit
import ctypes as C
import weakref

import numpy as np

from . import _capi


def device_count():
    """Number of GPUs this process can see (origin_device_count)."""
    n = C.c_int(0)
    _capi.call("origin_device_count", C.byref(n))
    return n.value


# Contexts still open when the interpreter exits are closed from an atexit handler: a context that
# survives into the HIP runtime's own static destructors (a reference cycle keeps it from __del__)
# leaves streams behind -- the CU-masked side stream among them -- and the runtime aborts the
# process there ("std::get: wrong index for variant"), after all the work is done but with a
# non-zero exit status.
_live_contexts = weakref.WeakSet()


@atexit.register
def _close_live_contexts():
    for ctx in list(_live_contexts):
        try:
            ctx.close()
        except Exception:
            pass


class Context:
    """One GPU, one stream (include/origin_hip.h: origin_ctx)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _capi.call("origin_ctx_create", int(device), C.byref(self._h))
        self.device = int(device)
        _live_contexts.add(self)

    # -- lifetime ----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _capi.load().origin_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def name(self):
        buf = C.create_string_buffer(256)
        _capi.call("origin_device_name", self._h, buf, 256)
        return buf.value.decode()

    def mem_info(self):
        f, t = C.c_size_t(), C.c_size_t()
        _capi.call("origin_mem_info", self._h, C.byref(f), C.byref(t))
        return f.value, t.value

    def aux_join(self):
        """Main stream waits for the work pending on the auxiliary stream (no host sync)."""
        _capi.call("origin_aux_join", self.handle)

    def sync(self):
        _capi.call("origin_sync", self._h)

    # -- greedy PCA tail hook (include/origin_hip.h origin_pca_set_tail_hook) -----------------
    _TAIL_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_int))

    def set_pca_tail_hook(self, fn, max_active=2):
        """``fn(areas)`` is called once per greedy-PCA run, from inside it, when at most
        ``max_active`` areas still iterate and the others have been written to the output;
        ``areas`` = indices of the areas that go on.  ``fn=None`` removes the hook."""
        if fn is None:
            self._tail_cb = None
            _capi.call("origin_pca_set_tail_hook", self._h, None, None, 0)
            return
        self._tail_exc = None

        def cb(_user, n, areas):
            try:
                fn([int(areas[i]) for i in range(n)])
            except BaseException as e:   # (an exception cannot cross the C frames)
                self._tail_exc = e
        self._tail_cb = self._TAIL_CB(cb)
        _capi.call("origin_pca_set_tail_hook", self._h, C.cast(self._tail_cb, C.c_void_p), None,
                   int(max_active))

    def pop_tail_hook_error(self):
        e, self._tail_exc = getattr(self, "_tail_exc", None), None
        return e

    def stream_ptr(self):
        s = C.c_void_p()
        _capi.call("origin_stream", self._h, C.byref(s))
        return s.value or 0

    # -- timers (HIP events on the context's stream) ---------------------------
    def timer_start(self, slot):
        _capi.call("origin_timer_start", self._h, slot)

    def timer_stop(self, slot):
        _capi.call("origin_timer_stop", self._h, slot)

    def timer_ms(self, slot):
        ms = C.c_float()
        _capi.call("origin_timer_ms", self._h, slot, C.byref(ms))
        return ms.value

    # -- built-in per-kernel-class profiler (HIP events on the stream) --------
    def prof_enable(self, level=1):
        """0 / False: off.  1 / True: one HIP-event pair per large kernel and per greedy-PCA run.
        2: every PCA kernel as well (an event pair costs ~10 us of stream time: ~5 ms per step)."""
        _capi.call("origin_prof_enable", self._h, int(level))

    def prof_reset(self):
        _capi.call("origin_prof_reset", self._h)

    def prof_report(self):
        """{kernel class: (total_ms, launches)} since the last reset (synchronises)."""
        out = {}
        for i in range(_capi.load().origin_prof_count()):
            name, ms, n = C.c_char_p(), C.c_double(), C.c_long()
            _capi.call("origin_prof_get", self._h, i, C.byref(name), C.byref(ms), C.byref(n))
            if n.value:
                out[name.value.decode()] = (ms.value, n.value)
        return out

    # -- arrays ------------------------------------------------------------
    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype):
        a = DeviceArray(self, shape, dtype)
        a.fill_bytes(0)
        return a

    def to_device(self, host, dtype=None):
        if (dtype is not None and np.dtype(dtype) == np.float32 and isinstance(host, np.ndarray)
                and host.dtype == np.float64 and host.flags.c_contiguous and host.size):
            # float64 in, float32 on the device: narrowed natively on the host pool into pinned
            # staging (origin_h2d_f64_as_f32) instead of a single-threaded astype + pageable copy
            a = DeviceArray(self, host.shape, np.float32)
            _capi.call("origin_h2d_f64_as_f32", self.handle, a.p, host.ctypes.data_as(C.c_void_p),
                       host.size)
            return a
        if isinstance(host, np.ndarray) and host.dtype == np.bool_ and dtype is not None \
                and np.dtype(dtype) == np.uint8:
            host = host.view(np.uint8)   # (no copy: bool is one byte of 0 / 1)
        host = np.ascontiguousarray(host, dtype=dtype)
        a = DeviceArray(self, host.shape, host.dtype)
        a.upload(host)
        return a


class DeviceArray:
    """A C-contiguous array in HBM owned by the library's allocator."""

    def __init__(self, ctx, shape, dtype, ptr=None, owner=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if np.iterable(shape) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        self._owner = owner
        if ptr is None:
            p = C.c_void_p()
            _capi.call("origin_malloc", ctx.handle, max(self.nbytes, 16), C.byref(p))
            self.ptr = p.value
            self._owned = True
        else:
            self.ptr = int(ptr)
            self._owned = False

    def free(self):
        if self._owned and self.ptr:
            try:
                _capi.call("origin_free", self.ctx.handle, C.c_void_p(self.ptr))
            finally:
                self.ptr = 0
                self._owned = False

    def __del__(self):
        try:
            if self._owned and self.ptr and self.ctx.handle.value:
                self.free()
        except Exception:
            pass

    @property
    def p(self):
        return C.c_void_p(self.ptr)

    def view(self, offset_elems, shape, dtype=None):
        """A non-owning view starting `offset_elems` elements into this array."""
        dtype = self.dtype if dtype is None else np.dtype(dtype)
        return DeviceArray(self.ctx, shape, dtype,
                           ptr=self.ptr + int(offset_elems) * self.dtype.itemsize, owner=self)

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and np.iterable(shape[0]) else shape
        v = DeviceArray(self.ctx, shape, self.dtype, ptr=self.ptr, owner=self)
        assert v.size == self.size
        return v

    def fill_bytes(self, byte):
        _capi.call("origin_memset", self.ctx.handle, self.p, int(byte), self.nbytes)

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size != self.size:
            raise ValueError(f"upload of {host.shape} into {self.shape}")
        _capi.call("origin_h2d", self.ctx.handle, self.p, host.ctypes.data_as(C.c_void_p),
                   self.nbytes)
        return self

    def to_host(self, out=None):
        if out is None:
            out = np.empty(self.shape, dtype=self.dtype)
        assert out.flags.c_contiguous and out.nbytes == self.nbytes
        _capi.call("origin_d2h", self.ctx.handle, out.ctypes.data_as(C.c_void_p), self.p,
                   self.nbytes)
        return out

    def to_host_f64(self, out=None):
        """Host float64 copy of a float32 (or float64) device array: what the reference's
        interface hands on.  float32 arrays are widened natively (pinned staging + host worker
        pool, ``origin_d2h_f32_as_f64``) instead of ``to_host().astype(float64)``."""
        if self.dtype == np.float64:
            return self.to_host(out)
        if self.dtype != np.float32:
            return self.to_host().astype(np.float64)
        if out is None:
            out = np.empty(self.shape, dtype=np.float64)
        assert out.flags.c_contiguous and out.size == self.size and out.dtype == np.float64
        _capi.call("origin_d2h_f32_as_f64", self.ctx.handle, out.ctypes.data_as(C.c_void_p), self.p,
                   self.size)
        return out

    def window(self, y0, y1, x0, x1, z0=0, z1=None):
        """Host copy of the box [z0:z1, y0:y1, x0:x1] of a (Nz, Ny, Nx) array (one strided
        device->host copy: nothing but the box crosses PCIe)."""
        Nz, Ny, Nx = self.shape
        z1 = Nz if z1 is None else z1
        y0, y1, x0, x1, z0, z1 = (int(v) for v in (y0, y1, x0, x1, z0, z1))
        if not (0 <= z0 < z1 <= Nz and 0 <= y0 < y1 <= Ny and 0 <= x0 < x1 <= Nx):
            raise ValueError(f"box [{z0}:{z1}, {y0}:{y1}, {x0}:{x1}] outside {self.shape}")
        nz, ny, nx = z1 - z0, y1 - y0, x1 - x0
        out = np.empty((nz, ny, nx), dtype=self.dtype)
        es = self.dtype.itemsize
        src = self.ptr + ((z0 * Ny + y0) * Nx + x0) * es
        _capi.call("origin_copy_box", self.ctx.handle, 1, out.ctypes.data_as(C.c_void_p), nx,
                   ny * nx, C.c_void_p(src), Nx, Ny * Nx, nz, ny, nx, es)
        return out

    def copy_from(self, other):
        assert other.nbytes == self.nbytes
        _capi.call("origin_d2d", self.ctx.handle, self.p, other.p, self.nbytes)
        return self

    def copy(self):
        return DeviceArray(self.ctx, self.shape, self.dtype).copy_from(self)

    # interop: any consumer of the array-interface protocol can view the buffer without a copy
    @property
    def __cuda_array_interface__(self):
        return dict(shape=self.shape, typestr=self.dtype.str, data=(self.ptr, False), version=3,
                    strides=None)


_default_ctx = {}


def default_context(device=0):
    """Process-wide context per device (the reference calls each step synchronously from one
    thread)."""
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx
