"""ctypes face of oracle/warp_ref.c (TEST INFRASTRUCTURE ONLY: the plain-C restatement of the bit-exact index arithmetic).
Builds oracle/libwarp_ref.so with gcc on first use if `make -C oracle` has not been run."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libwarp_ref.so")
        src = os.path.join(_HERE, "warp_ref.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(so)
        f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
        _lib.warp_ref_up4.argtypes = [f32p, ctypes.c_int, ctypes.c_int, f32p, ctypes.c_float, ctypes.c_float, ctypes.c_float]
        _lib.warp_ref_corners.argtypes = [f32p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p, f32p]
        _lib.warp_ref_sample.argtypes = [f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p, f32p, ctypes.c_long, f32p]
    return _lib


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def up4(plane, pre=1.0, post_a=1.0, post_b=0.0):
    """plane (h, w) float32 -> (4h, 4w)"""
    plane = np.ascontiguousarray(plane, dtype=np.float32)
    h, w = plane.shape
    out = np.empty((4 * h, 4 * w), dtype=np.float32)
    lib().warp_ref_up4(_f(plane), h, w, _f(out), pre, post_a, post_b)
    return out


def corners(grid, IH, IW, half_grid=False):
    """grid (..., 2) float32 -> (corner (..., 2) int32 = floor x0,y0 clamped to [-2, size+1]; weights (..., 4))"""
    grid = np.ascontiguousarray(grid, dtype=np.float32)
    n = grid.size // 2
    c = np.empty(grid.shape, dtype=np.int32)
    wts = np.empty(grid.shape[:-1] + (4,), dtype=np.float32)
    lib().warp_ref_corners(_f(grid), n, IH, IW, int(half_grid), _i(c), _f(wts))
    return c, wts


def warp(img, grid, half_grid=False):
    """img (C, IH, IW), grid (H, W, 2) -> (C, H, W): F.grid_sample(bilinear, zeros, align_corners=False)"""
    img = np.ascontiguousarray(img, dtype=np.float32)
    C_, IH, IW = img.shape
    c, wts = corners(grid, IH, IW, half_grid)
    n = c.size // 2
    out = np.empty((C_,) + grid.shape[:-1], dtype=np.float32)
    lib().warp_ref_sample(_f(img), C_, IH, IW, _i(c), _f(wts), n, _f(out))
    return out
