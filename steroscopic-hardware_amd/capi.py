"""ctypes binding of the C ABI in include/sadhip.h (lib/libsadhip.so).

This is the Python counterpart of the cgo stub in INTEGRATION.md: same entry points, same
argument meaning.  numpy arrays travel as host pointers; torch CUDA tensors travel as raw
device pointers (``tensor.data_ptr()``) into the ``*_dev_*`` entry point -- torch is plumbing
for device memory and streams only.

No fallback: if the shared library is missing or fails to load, importing this module's
``lib()`` raises; if there is no GPU, every compute call raises SadHipError.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsadhip.so")

OK = 0
ERR_NULL, ERR_DIMS, ERR_STRIDE, ERR_BLOCK, ERR_MAXDISP, ERR_REGION, ERR_BATCH, ERR_NO_DEVICE, ERR_HIP, ERR_ARG, ERR_NOMEM = range(1, 12)
FLAG_RAW = 1
FLAG_FORCE_GENERIC = 2
FLAG_MIRROR_SADC = 4
FLAG_MIRROR_TESTPY = 8
FLAG_MIRROR_VERILOG = 16
FLAG_HOST_GRAY = 32           # sadhip_disparity_rgba_u8 only: the copy threads write luminance rows (a quarter of the bytes uploaded)
KIND_GENERIC, KIND_COLWAVE, KIND_MIRROR_SADC, KIND_MIRROR_TESTPY, KIND_MIRROR_VERILOG, KIND_GENERIC_SLIDING = range(6)
GENERIC_MAX_WORK = 65536
ABI_VERSION = 5
MAX_BLOCK = 255
MAX_MAXDISP = 4096

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)

# every symbol include/sadhip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sadhip_abi_version": (C.c_int, []),
    "sadhip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sadhip_status_string": (C.c_char_p, [C.c_int]),
    "sadhip_last_error": (C.c_char_p, []),
    "sadhip_ctx_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sadhip_ctx_destroy": (C.c_int, [C.c_void_p]),
    "sadhip_ctx_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "sadhip_ctx_set_copy_threads": (C.c_int, [C.c_void_p, C.c_int]),
    "sadhip_ctx_get_copy_threads": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "sadhip_ctx_set_copy_spin_us": (C.c_int, [C.c_void_p, C.c_int]),
    "sadhip_ctx_get_copy_spin_us": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "sadhip_disparity_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_int]),
    "sadhip_disparity_region_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sadhip_disparity_lr_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "sadhip_disparity_batch_u8": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint,
                                            C.POINTER(C.c_void_p), C.c_int]),
    "sadhip_disparity_dev_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint,
                                          C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "sadhip_sad_windows_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sadhip_gray_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "sadhip_disparity_rgba_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_int]),
    "sadhip_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sadhip_kernel_kind": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]),
    "sadhip_last_kernel_kind": (C.c_int, []),
    "sadhip_plan_describe": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "sadhip_plan_describe_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "sadhip_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "sadhip_host_free": (C.c_int, [C.c_void_p]),
    "sadhip_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "sadhip_host_unregister": (C.c_int, [C.c_void_p]),
    "sadhip_pool_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sadhip_pool_destroy": (C.c_int, [C.c_void_p]),
    "sadhip_pool_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "sadhip_pool_set_copy_threads": (C.c_int, [C.c_void_p, C.c_int]),
    "sadhip_pool_set_copy_spin_us": (C.c_int, [C.c_void_p, C.c_int]),
    "sadhip_pool_batch_u8": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint,
                                       C.POINTER(C.c_void_p), C.c_int]),
    "sadhip_pool_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sadhip_pool_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "sadhip_device_describe": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "sadhip_pool_host_alloc": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
}

_LIB = None


class SadHipError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__(f"sadhip status {status}: {detail}")


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of lib/libsadhip.so (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", csrc], stdout=subprocess.DEVNULL)
    return LIB_PATH


def _preload_hip_runtime():
    """libsadhip.so is linked without a DT_NEEDED on the HIP runtime (csrc/Makefile): make ONE
    libamdhip64 global before loading it.  Inside a PyTorch process that must be the copy torch
    bundles (torch/lib/libamdhip64.so); otherwise the system ROCm's.  SADHIP_HIP_RUNTIME overrides."""
    cands = []
    if os.environ.get("SADHIP_HIP_RUNTIME"):
        cands.append(os.environ["SADHIP_HIP_RUNTIME"])
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            import torch  # noqa: F401  (loads its bundled runtime first, with torch's own rpath)
            cands.append(os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so"))
    except Exception:
        pass
    cands += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    errs = []
    for c in cands:
        if os.path.sep in c and not os.path.exists(c):
            continue
        try:
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
        except OSError as e:
            errs.append(f"{c}: {e}")
    raise ImportError("no HIP runtime (libamdhip64) could be loaded: " + "; ".join(errs))


def lib():
    """Loads libsadhip.so; raises (never falls back) when it is absent or lacks a symbol."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        _preload_hip_runtime()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)          # AttributeError if the ABI symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if l.sadhip_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH}: ABI version {l.sadhip_abi_version()}, this binding needs {ABI_VERSION}: rebuild it")
        _LIB = l
    return _LIB


def _check(status):
    if status != OK:
        raise SadHipError(status, (lib().sadhip_last_error() or b"").decode(errors="replace"))


def device_count():
    n = C.c_int(0)
    _check(lib().sadhip_device_count(C.byref(n)))
    return n.value


def device_describe(device=0):
    """Where a device hangs on the host (sadhip_device_describe): dict device, pci, numa_node, cpus (a set of CPU numbers:
    the device's local CPUs within this process's affinity mask; empty when unknown)."""
    buf = C.create_string_buffer(1024)
    _check(lib().sadhip_device_describe(device, buf, len(buf)))
    d = dict(f.split("=", 1) for f in buf.value.decode().split())
    cpus = set()
    if d.get("cpus", "-") != "-":
        for part in d["cpus"].split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
    return {"device": int(d["device"]), "pci": d["pci"], "numa_node": int(d["numa_node"]), "cpus": cpus}


def kernel_kind(w, h, block, maxdisp, flags=0):
    return lib().sadhip_kernel_kind(w, h, block, maxdisp, flags)


def last_kernel_kind():
    """Kernel the calling thread's last successful disparity call actually launched (KIND_*; -1: none)."""
    return lib().sadhip_last_kernel_kind()


def plan_describe(w, h, block, maxdisp):
    """The passes the column-wave kernel would run (no GPU needed): list of dicts with the integer
    fields dbase, count, p, nd, nw, packed, extra, w16, nwh, lds; [] when the generic kernel would run."""
    buf = C.create_string_buffer(16384)
    n = lib().sadhip_plan_describe(w, h, block, maxdisp, buf, len(buf))
    if n < 0:
        raise SadHipError(ERR_ARG, "invalid parameters")
    return [{k: int(v) for k, v in (f.split("=") for f in line.split())} for line in buf.value.decode().splitlines()]


def plan_describe_batch(w, h, block, maxdisp, n):
    """plan_describe for a batch of n frames on the current HIP device, with the launch geometry of every
    pass: tiles, strips (per frame), rows (output rows per strip), wg_per_cu, device (1 = occupancy asked from
    the runtime).  Strip s of a frame computes output rows [s*rows, min((s+1)*rows, h))."""
    buf = C.create_string_buffer(16384)
    k = lib().sadhip_plan_describe_batch(w, h, block, maxdisp, n, buf, len(buf))
    if k < 0:
        raise SadHipError(ERR_ARG, "invalid parameters")
    return [{k: int(v) for k, v in (f.split("=") for f in line.split())} for line in buf.value.decode().splitlines()]


class HostBuffer:
    """Page-locked host memory from sadhip_host_alloc, exposed as a uint8 numpy array (``.array``, any shape
    with the same number of bytes via ``.view(shape)``).  Frames kept in it travel with asynchronous linear
    copies (include/sadhip.h "page-locked host memory").  close() frees it: do not use views afterwards."""

    def __init__(self, nbytes):
        self._p = C.c_void_p()
        _check(lib().sadhip_host_alloc(nbytes, C.byref(self._p)))
        self.nbytes = nbytes
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self._p.value))

    @classmethod
    def _adopt(cls, p, nbytes):
        """Wraps page-locked memory the library has already allocated (sadhip_pool_host_alloc)."""
        self = cls.__new__(cls)
        self._p, self.nbytes = p, nbytes
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p.value))
        return self

    def view(self, shape):
        return self.array[:int(np.prod(shape))].reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            _check(lib().sadhip_host_free(self._p))
            self._p = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MappedBuffer:
    """Memory the CALLER owns, page-locked in place (sadhip_host_register): an anonymous mmap of whole 4 KiB pages -- the only
    kind of memory a registration accepts (ABI 5: page-aligned whole pages of a mapping of the caller's own, never the
    program break; include/sadhip.h has the contract).  ``.array`` / ``.view(shape)`` like HostBuffer; close() unregisters
    first and unmaps afterwards."""

    def __init__(self, nbytes):
        import mmap
        self.nbytes = (int(nbytes) + 4095) // 4096 * 4096
        self._map = mmap.mmap(-1, self.nbytes)
        self.array = np.frombuffer(self._map, dtype=np.uint8)
        self._registered = False
        _check(lib().sadhip_host_register(self.array.ctypes.data, self.nbytes))
        self._registered = True

    def view(self, shape):
        return self.array[:int(np.prod(shape))].reshape(shape)

    def close(self):
        if self._registered:
            self._registered = False
            _check(lib().sadhip_host_unregister(self.array.ctypes.data))
        self.array = None                                     # (the mapping is released when the last view of it is)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _img(a, name):
    if not isinstance(a, np.ndarray) or a.dtype != np.uint8 or a.ndim != 2 or a.strides[1] != 1 or a.strides[0] < 0:
        raise TypeError(f"{name}: expected a 2-D uint8 numpy array with unit column stride")
    return a


class Context:
    """Per-device context (sadhip_ctx).  Use as a context manager or call close()."""

    def __init__(self, device=0, max_w=1920, max_h=1080, max_batch=1):
        self._h = C.c_void_p()
        _check(lib().sadhip_ctx_create(device, max_w, max_h, max_batch, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().sadhip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer entry points ---------------------------------------------------------
    def set_copy_threads(self, n):
        """Host threads (the caller included) that stage frames which are not in page-locked memory: 1..32."""
        _check(lib().sadhip_ctx_set_copy_threads(self._h, int(n)))

    def copy_threads(self):
        n = C.c_int(0)
        _check(lib().sadhip_ctx_get_copy_threads(self._h, C.byref(n)))
        return n.value

    def set_copy_spin_us(self, us):
        """How long a copy thread polls for its next piece before it sleeps: 0..10000 us (0 = sleep at once)."""
        _check(lib().sadhip_ctx_set_copy_spin_us(self._h, int(us)))

    def copy_spin_us(self):
        n = C.c_int(0)
        _check(lib().sadhip_ctx_get_copy_spin_us(self._h, C.byref(n)))
        return n.value

    def disparity(self, left, right, block, maxdisp, flags=0, out=None):
        _img(left, "left"), _img(right, "right")
        if left.shape != right.shape:
            raise SadHipError(ERR_DIMS, "left/right shape mismatch")
        h, w = left.shape
        if out is None:
            out = np.empty((h, w), np.uint8)
        _img(out, "out")
        _check(lib().sadhip_disparity_u8(self._h, left.ctypes.data, right.ctypes.data, w, h,
                                         left.strides[0], right.strides[0], block, maxdisp, flags,
                                         out.ctypes.data, out.strides[0]))
        return out

    def disparity_region(self, left, right, block, maxdisp, region, flags=0):
        _img(left, "left"), _img(right, "right")
        if left.shape != right.shape:
            raise SadHipError(ERR_DIMS, "left/right shape mismatch")
        h, w = left.shape
        x0, y0, x1, y1 = region
        out = np.empty((max(y1 - y0, 0), max(x1 - x0, 0)), np.uint8)
        _check(lib().sadhip_disparity_region_u8(self._h, left.ctypes.data, right.ctypes.data, w, h,
                                                left.strides[0], right.strides[0], block, maxdisp, flags,
                                                x0, y0, x1, y1, out.ctypes.data if out.size else np.empty(1, np.uint8).ctypes.data))
        return out

    def disparity_lr(self, left, right, block, maxdisp, region=None, flags=0):
        """Left and right images with their OWN sizes (sadhip_disparity_lr_u8): the reference's behaviour for
        independently decoded pictures (sad.go:212-218,225-227,231-233).  Region of the left image; region-local output."""
        _img(left, "left"), _img(right, "right")
        lh, lw = left.shape
        rh, rw = right.shape
        x0, y0, x1, y1 = region if region is not None else (0, 0, lw, lh)
        out = np.empty((max(y1 - y0, 0), max(x1 - x0, 0)), np.uint8)
        _check(lib().sadhip_disparity_lr_u8(self._h, left.ctypes.data, lw, lh, left.strides[0],
                                            right.ctypes.data, rw, rh, right.strides[0], block, maxdisp, flags,
                                            x0, y0, x1, y1, out.ctypes.data if out.size else np.empty(1, np.uint8).ctypes.data,
                                            max(x1 - x0, 1)))
        return out

    def disparity_batch(self, lefts, rights, block, maxdisp, flags=0, outs=None):
        n = len(lefts)
        assert len(rights) == n
        if n == 0:
            _check(lib().sadhip_disparity_batch_u8(self._h, 0, None, None, 1, 1, 1, 1, block, maxdisp, flags, None, 1))
            return []
        h, w = lefts[0].shape
        ls, rs = lefts[0].strides[0], rights[0].strides[0]
        for a in list(lefts) + list(rights):
            _img(a, "frame")
            assert a.shape == (h, w)
        assert all(a.strides[0] == ls for a in lefts) and all(a.strides[0] == rs for a in rights)
        if outs is None:
            outs = [np.empty((h, w), np.uint8) for _ in range(n)]
        os_ = outs[0].strides[0]                                     # one row stride for all outputs (>= w: column windows allowed)
        assert len(outs) == n and all(o.shape == (h, w) and o.strides == (os_, 1) for o in outs) and os_ >= w
        pa = (C.c_void_p * n)(*[a.ctypes.data for a in lefts])
        pb = (C.c_void_p * n)(*[a.ctypes.data for a in rights])
        po = (C.c_void_p * n)(*[a.ctypes.data for a in outs])
        _check(lib().sadhip_disparity_batch_u8(self._h, n, pa, pb, w, h, ls, rs, block, maxdisp, flags, po, os_))
        return outs

    def sad_windows(self, left, right, block, queries):
        _img(left, "left"), _img(right, "right")
        q = np.ascontiguousarray(np.asarray(queries, dtype=np.int32).reshape(-1, 4))
        out = np.zeros(len(q), np.int32)
        _check(lib().sadhip_sad_windows_u8(self._h, left.ctypes.data, left.shape[1], left.shape[0], left.strides[0],
                                           right.ctypes.data, right.shape[1], right.shape[0], right.strides[0],
                                           block, len(q), q.ctypes.data if len(q) else None,
                                           out.ctypes.data if len(q) else None))
        return out

    @staticmethod
    def _pixel_rows(pix):
        """A decoded picture as the ABI takes it: interleaved pixels, rows `strides[0]` bytes apart (image.NRGBA's Stride) -- a
        sub-picture view is passed as it is, anything else is made contiguous first."""
        if pix.ndim == 3 and pix.strides[2] == 1 and pix.strides[1] == pix.shape[2] and pix.strides[0] >= pix.shape[1] * pix.shape[2]:
            return pix
        return np.ascontiguousarray(pix)

    def gray(self, pix):
        if not isinstance(pix, np.ndarray) or pix.dtype != np.uint8 or pix.ndim != 3 or pix.shape[2] not in (3, 4):
            raise TypeError("pix: expected HxWx3 or HxWx4 uint8")
        pix = self._pixel_rows(pix)
        h, w, c = pix.shape
        g = np.empty((h, w), np.uint8)
        _check(lib().sadhip_gray_u8(self._h, pix.ctypes.data, c, w, h, pix.strides[0], g.ctypes.data, g.strides[0]))
        return g

    def disparity_rgba(self, left_pix, right_pix, block, maxdisp, flags=0):
        assert left_pix.shape == right_pix.shape and left_pix.dtype == np.uint8 and left_pix.ndim == 3
        left_pix = self._pixel_rows(left_pix); right_pix = self._pixel_rows(right_pix)
        h, w, c = left_pix.shape
        out = np.empty((h, w), np.uint8)
        _check(lib().sadhip_disparity_rgba_u8(self._h, left_pix.ctypes.data, right_pix.ctypes.data, c, w, h,
                                              left_pix.strides[0], right_pix.strides[0], block, maxdisp, flags,
                                              out.ctypes.data, out.strides[0]))
        return out

    def last_timing(self):
        k, t = C.c_float(0), C.c_float(0)
        _check(lib().sadhip_last_timing(self._h, C.byref(k), C.byref(t)))
        return k.value, t.value

    # ---- device-resident entry point (torch CUDA tensors as raw pointers) --------------------
    def disparity_dev(self, d_left, d_right, d_out, block, maxdisp, flags=0, stream=None):
        """d_left/d_right/d_out: torch uint8 CUDA tensors of shape [n,h,w] (or [h,w]), row-contiguous.
        Enqueues on ``stream`` (an int hipStream_t; default: torch's current stream) and returns."""
        import torch
        if d_left.dim() == 2:
            d_left, d_right, d_out = d_left[None], d_right[None], d_out[None]
        n, h, w = d_left.shape
        for t in (d_left, d_right, d_out):
            assert t.is_cuda and t.dtype == torch.uint8 and t.shape == (n, h, w) and t.stride(2) == 1
        if stream is None:
            stream = torch.cuda.current_stream(d_left.device).cuda_stream
        _check(lib().sadhip_disparity_dev_u8(self._h, n, d_left.data_ptr(), d_right.data_ptr(),
                                             d_left.stride(0), d_right.stride(0), w, h,
                                             d_left.stride(1), d_right.stride(1), block, maxdisp, flags,
                                             d_out.data_ptr(), d_out.stride(0), d_out.stride(1), stream))
        return d_out


def _frames(lefts, rights, outs):
    """Checks one batch of equally shaped u8 frames; returns (n, w, h, ls, rs, outs, pointer arrays)."""
    n = len(lefts)
    assert len(rights) == n
    h, w = lefts[0].shape
    ls, rs = lefts[0].strides[0], rights[0].strides[0]
    for a in list(lefts) + list(rights):
        _img(a, "frame")
        assert a.shape == (h, w)
    assert all(a.strides[0] == ls for a in lefts) and all(a.strides[0] == rs for a in rights)
    if outs is None:
        outs = [np.empty((h, w), np.uint8) for _ in range(n)]
    assert len(outs) == n and all(o.shape == (h, w) and o.strides == (w, 1) for o in outs)
    pa = (C.c_void_p * n)(*[a.ctypes.data for a in lefts])
    pb = (C.c_void_p * n)(*[a.ctypes.data for a in rights])
    po = (C.c_void_p * n)(*[a.ctypes.data for a in outs])
    return n, w, h, ls, rs, outs, pa, pb, po


class Pool:
    """sadhip_pool: one context + host thread per entry of ``devices`` (an ordinal may repeat);
    frame i of a batch runs on entry i mod len(devices)."""

    def __init__(self, devices, max_w=1920, max_h=1080):
        devices = list(devices)
        self._h = C.c_void_p()
        arr = (C.c_int * max(len(devices), 1))(*devices)
        _check(lib().sadhip_pool_create(arr, len(devices), max_w, max_h, C.byref(self._h)))
        self.devices = devices

    def close(self):
        if self._h:
            lib().sadhip_pool_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        n = C.c_int(0)
        _check(lib().sadhip_pool_size(self._h, C.byref(n)))
        return n.value

    def set_copy_threads(self, n):
        """Host threads per entry (its worker included) that stage frames which are not in page-locked memory: 1..32."""
        _check(lib().sadhip_pool_set_copy_threads(self._h, int(n)))

    def set_copy_spin_us(self, us):
        _check(lib().sadhip_pool_set_copy_spin_us(self._h, int(us)))

    def batch(self, lefts, rights, block, maxdisp, flags=0, outs=None):
        if len(lefts) == 0:
            _check(lib().sadhip_pool_batch_u8(self._h, 0, None, None, 1, 1, 1, 1, block, maxdisp, flags, None, 1))
            return []
        n, w, h, ls, rs, outs, pa, pb, po = _frames(lefts, rights, outs)
        _check(lib().sadhip_pool_batch_u8(self._h, n, pa, pb, w, h, ls, rs, block, maxdisp, flags, po, w))
        return outs

    def host_alloc(self, entry, nbytes):
        """Page-locked memory allocated and first-touched by entry's worker thread (on the CPUs next to its device): a
        HostBuffer (close() frees it)."""
        p = C.c_void_p()
        _check(lib().sadhip_pool_host_alloc(self._h, entry, nbytes, C.byref(p)))
        return HostBuffer._adopt(p, nbytes)

    def describe(self):
        """Host placement of the pool's workers (sadhip_pool_describe): list of dicts entry, device, pci, numa_node,
        pinned (worker thread pinned to the CPUs of its device's NUMA node), cpus."""
        buf = C.create_string_buffer(8192)
        _check(lib().sadhip_pool_describe(self._h, buf, len(buf)))
        out = []
        for line in buf.value.decode().splitlines():
            d = dict(f.split("=", 1) for f in line.split())
            out.append({k: (int(v) if k in ("entry", "device", "numa_node", "pinned") else v) for k, v in d.items()})
        return out

    def last_timing(self):
        k, t = C.c_float(0), C.c_float(0)
        _check(lib().sadhip_pool_last_timing(self._h, C.byref(k), C.byref(t)))
        return k.value, t.value
