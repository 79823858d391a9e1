"""Seeded synthetic pages / frames with ground truth.

Thin ctypes wrapper over csrc/synth.cpp (host-only).  Not on the product path.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

PAGE_SEED = 0x511DE0
FRAME_SEED = 0xF4A3E5

_lib = None


def _L():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build_synth())
    return _lib


def pages(n, w=2001, h=1125, seed=PAGE_SEED, threads=None):
    """n pages, uint8 [n, h, w, 3] BGR."""
    out = np.empty((n, h, w, 3), np.uint8)
    threads = threads or min(os.cpu_count() or 1, 64)
    _L().slideo_synth_pages(C.c_uint64(seed), n, w, h, out.ctypes.data_as(C.c_void_p), threads)
    return out


def frames(page_stack, n, w=1920, h=1080, first=0, seed=FRAME_SEED, threads=None):
    """n frames showing random pages of `page_stack` ([P, ph, pw, 3]).

    Returns (frames [n,h,w,3], truth_page [n] (-1 = no slide), truth_M [n,2,3] slide->frame).
    """
    page_stack = np.ascontiguousarray(page_stack, np.uint8)
    P, ph, pw, _ = page_stack.shape
    out = np.empty((n, h, w, 3), np.uint8)
    tp = np.empty(n, np.int32)
    tm = np.empty((n, 6), np.float64)
    threads = threads or min(os.cpu_count() or 1, 64)
    _L().slideo_synth_frames(C.c_uint64(seed), C.c_int64(first), n,
                             page_stack.ctypes.data_as(C.c_void_p), P, pw, ph, w, h,
                             out.ctypes.data_as(C.c_void_p), tp.ctypes.data_as(C.c_void_p),
                             tm.ctypes.data_as(C.c_void_p), threads)
    return out, tp, tm.reshape(n, 2, 3)


def frames_persp(page_stack, n, w=1920, h=1080, persp=0.15, first=0, seed=FRAME_SEED, threads=None):
    """n frames showing random pages under a HOMOGRAPHY (keystone of about persp / 2 across the slide, then the
    similarity of frames()).  Returns (frames, truth_page, truth_H [n,3,3] slide->frame)."""
    page_stack = np.ascontiguousarray(page_stack, np.uint8)
    P, ph, pw, _ = page_stack.shape
    out = np.empty((n, h, w, 3), np.uint8)
    tp = np.empty(n, np.int32)
    th = np.empty((n, 9), np.float64)
    threads = threads or min(os.cpu_count() or 1, 64)
    _L().slideo_synth_frames_persp(C.c_uint64(seed), C.c_int64(first), n,
                                   page_stack.ctypes.data_as(C.c_void_p), P, pw, ph, w, h, C.c_double(persp),
                                   out.ctypes.data_as(C.c_void_p), tp.ctypes.data_as(C.c_void_p),
                                   th.ctypes.data_as(C.c_void_p), threads)
    return out, tp, th.reshape(n, 3, 3)
