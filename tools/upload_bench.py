#!/usr/bin/env python3
"""How fast does a pageable N x N matrix reach the device as one copy, and as block-row 2-D copies of its lower
trapezoids (what the dense LCP needs first) followed by the rest?"""
import ctypes
import time

import numpy as np

hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipMemcpy2DAsync.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                 ctypes.c_int, ctypes.c_void_p]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
H2D = 1
N = 2048
A = np.random.default_rng(0).uniform(-1, 1, (N, N))
d = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(d), N * N * 8) == 0
src = A.ctypes.data


def full():
    hip.hipMemcpyAsync(d, src, N * N * 8, H2D, None)
    hip.hipStreamSynchronize(None)


def chunks(rows):
    t_low = None
    t0 = time.perf_counter()
    for r0 in range(0, N, rows):
        r1 = r0 + rows
        hip.hipMemcpy2DAsync(ctypes.c_void_p(d.value + r0 * N * 8), N * 8, ctypes.c_void_p(src + r0 * N * 8), N * 8, r1 * 8, rows, H2D, None)
    hip.hipStreamSynchronize(None)
    t_low = time.perf_counter() - t0
    for r0 in range(0, N - rows, rows):
        r1 = r0 + rows
        hip.hipMemcpy2DAsync(ctypes.c_void_p(d.value + r0 * N * 8 + r1 * 8), N * 8, ctypes.c_void_p(src + r0 * N * 8 + r1 * 8), N * 8,
                             (N - r1) * 8, rows, H2D, None)
    hip.hipStreamSynchronize(None)
    return t_low, time.perf_counter() - t0


for _ in range(3):
    full()
best = min((lambda t0: (full(), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(5))
print("one copy: %.3f ms" % (best * 1e3))
for rows in (64, 128, 256, 512):
    for _ in range(2):
        chunks(rows)
    res = [chunks(rows) for _ in range(5)]
    print("block rows of %d: lower part %.3f ms, everything %.3f ms" % (rows, min(r[0] for r in res) * 1e3, min(r[1] for r in res) * 1e3))
