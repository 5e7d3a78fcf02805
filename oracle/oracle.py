"""ctypes loader for the parity oracle (oracle/despair_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    so = os.path.join(_HERE, "libdespair_oracle.so")
    src = os.path.join(_HERE, "despair_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libdespair_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.oracle_sum_absolute_differences.restype = C.c_int
        _LIB.oracle_region_literal.restype = C.c_int
        _LIB.oracle_region_literal_lr.restype = C.c_int
        _LIB.oracle_region_closed.restype = C.c_int
        _LIB.oracle_rows_literal_mt.restype = C.c_int
        _LIB.oracle_mirror_verilog_strip.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(_u8p)


def _chk(a):
    assert a.dtype == np.uint8 and a.ndim == 2 and a.strides[1] == 1
    return a


def sum_absolute_differences(left, right, leftX, leftY, rightX, rightY, blockSize):
    """pkg/despair/sad.go:205-244 on numpy u8 images (right may differ in size)."""
    _chk(left), _chk(right)
    return lib().oracle_sum_absolute_differences(
        _p(left), left.shape[1], left.shape[0], left.strides[0],
        _p(right), right.shape[1], right.shape[0], right.strides[0],
        leftX, leftY, rightX, rightY, blockSize)


def _region(fn, left, right, block, maxdisp, region, raw):
    _chk(left), _chk(right)
    assert left.shape == right.shape
    h, w = left.shape
    x0, y0, x1, y1 = region if region is not None else (0, 0, w, h)
    out = np.zeros((max(y1 - y0, 0), max(x1 - x0, 0)), np.uint8)
    rc = fn(_p(left), _p(right), w, h, left.strides[0], right.strides[0], block, maxdisp,
            x0, y0, x1, y1, _p(out), 1 if raw else 0)
    if rc != 0:
        raise ZeroDivisionError("MaxDisparity <= 0 (the reference panics: integer divide by zero, sad.go:92)")
    return out


def disparity_literal(left, right, block, maxdisp, region=None, raw=False):
    """Literal restatement of sad.go:55-94 over region (x0,y0,x1,y1); region-local output."""
    return _region(lib().oracle_region_literal, left, right, block, maxdisp, region, raw)


def disparity_literal_lr(left, right, block, maxdisp, region=None, raw=False):
    """The same literal loops for a right image of ANY size (sad.go:212-218,225-227,231-233 clamp / break per
    image); region of the left image, region-local output."""
    _chk(left), _chk(right)
    lh, lw = left.shape
    rh, rw = right.shape
    x0, y0, x1, y1 = region if region is not None else (0, 0, lw, lh)
    out = np.zeros((max(y1 - y0, 0), max(x1 - x0, 0)), np.uint8)
    rc = lib().oracle_region_literal_lr(_p(left), lw, lh, left.strides[0], _p(right), rw, rh, right.strides[0],
                                        block, maxdisp, x0, y0, x1, y1, _p(out), 1 if raw else 0)
    if rc != 0:
        raise ZeroDivisionError("MaxDisparity <= 0 (the reference panics: integer divide by zero, sad.go:92)")
    return out


def disparity_closed(left, right, block, maxdisp, region=None, raw=False):
    """Closed form of the same function."""
    return _region(lib().oracle_region_closed, left, right, block, maxdisp, region, raw)


def disparity_rows_mt(left, right, block, maxdisp, ya, yb, nthreads):
    """Literal restatement, rows [ya,yb), worker threads over output.go-style row bands."""
    _chk(left), _chk(right)
    h, w = left.shape
    out = np.zeros((yb - ya, w), np.uint8)
    rc = lib().oracle_rows_literal_mt(_p(left), _p(right), w, h, left.strides[0], right.strides[0],
                                      block, maxdisp, ya, yb, _p(out), nthreads)
    assert rc == 0
    return out


def gray_from_nrgba(rgba):
    assert rgba.dtype == np.uint8 and rgba.ndim == 3 and rgba.shape[2] == 4
    rgba = np.ascontiguousarray(rgba)
    h, w, _ = rgba.shape
    g = np.zeros((h, w), np.uint8)
    lib().oracle_gray_from_nrgba(_p(rgba), w, h, rgba.strides[0], _p(g), g.strides[0])
    return g


def gray_from_rgb8(pix, refbug=False):
    assert pix.dtype == np.uint8 and pix.ndim == 3 and pix.shape[2] in (3, 4)
    pix = np.ascontiguousarray(pix)
    h, w, c = pix.shape
    g = np.zeros((h, w), np.uint8)
    fn = lib().oracle_gray_from_rgba8_refbug if refbug else lib().oracle_gray_from_rgb8
    fn(_p(pix), w, h, pix.strides[0], c, _p(g), g.strides[0])
    return g


def mirror_sadc(L, R, win, maxd):
    L = np.ascontiguousarray(L); R = np.ascontiguousarray(R)
    out = np.zeros_like(L)
    lib().oracle_mirror_sadc(_p(L), _p(R), L.shape[1], L.shape[0], win, maxd, _p(out))
    return out


def mirror_testpy(L, R, win, maxd):
    L = np.ascontiguousarray(L); R = np.ascontiguousarray(R)
    out = np.zeros_like(L)
    lib().oracle_mirror_testpy(_p(L), _p(R), L.shape[1], L.shape[0], win, maxd, _p(out))
    return out


def mirror_verilog_strip(L, R, win, ndisp):
    L = np.ascontiguousarray(L); R = np.ascontiguousarray(R)
    ms = C.c_int(0)
    d = lib().oracle_mirror_verilog_strip(_p(L), _p(R), win, L.shape[1], ndisp, C.byref(ms))
    return d, ms.value


def mirror_verilog(L, R, win, ndisp):
    """compute_SAD.v's strip search for every interior pixel (border 0), raw d."""
    L = np.ascontiguousarray(L); R = np.ascontiguousarray(R)
    out = np.zeros_like(L)
    lib().oracle_mirror_verilog(_p(L), _p(R), L.shape[1], L.shape[0], win, ndisp, _p(out))
    return out


def synth_pair(seed, w, h):
    L = np.zeros((h, w), np.uint8); R = np.zeros((h, w), np.uint8)
    lib().oracle_synth_pair(C.c_uint64(seed), w, h, _p(L), _p(R))
    return L, R
