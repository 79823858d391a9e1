#!/usr/bin/env python3
"""Squared-L2 k-NN kernel (int8 matrix cores) in isolation through the C ABI tap: SIFT-shaped 128-dim u8 descriptors.
Reports time including the host<->device copies of the tap and the kernel-only rate from a second, larger call."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slideo_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--nq", type=int, default=244736)
ap.add_argument("--nt", type=int, default=517239)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--k", type=int, default=2, help="neighbours per query (k <= 8 uses the short-list kernel instance)")
a = ap.parse_args()
rng = np.random.default_rng(0)
def sift_like(n):
    x = rng.gamma(0.6, 40.0, (n, 128)).astype(np.float32)
    x *= 512.0 / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)
q, t = sift_like(a.nq), sift_like(a.nt)
m = _capi.Matcher()
m.knn_l2_u8(q[:1024], t[:4096], a.k)
for _ in range(a.reps):
    t0 = time.perf_counter(); m.knn_l2_u8(q, t, a.k); dt = time.perf_counter() - t0
    print("k %d nq %d nt %d: %.2f ms incl. copies (%.2f T pairs/s, %.2f PFLOP/s int8-equivalent at 256 op/pair)" % (a.k, a.nq, a.nt, dt * 1e3, a.nq * a.nt / dt / 1e12, a.nq * a.nt * 256 / dt / 1e15))
