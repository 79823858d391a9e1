#!/usr/bin/env python3
"""kNN kernel in isolation (through the C ABI tap): random 256-bit descriptors."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slideo_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--nq", type=int, default=65536)
ap.add_argument("--nt", type=int, default=262144)
ap.add_argument("--engine", default="mfma")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
rng = np.random.default_rng(0)
q = rng.integers(0, 256, (a.nq, 32), dtype=np.uint8)
t = rng.integers(0, 256, (a.nt, 32), dtype=np.uint8)
m = _capi.Matcher()
m.set_knn_engine(a.engine)
m.knn(q[:1024], t[:4096], 30)
for _ in range(a.reps):
    t0 = time.perf_counter(); m.knn(q, t, 30); dt = time.perf_counter() - t0
    print("engine %s nq %d nt %d: %.2f ms incl. copies (%.2f T pairs/s)" % (a.engine, a.nq, a.nt, dt * 1e3, a.nq * a.nt / dt / 1e12))
