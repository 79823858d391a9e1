#!/opt/conda/bin/python3.9
"""Regenerates tests/golden/crosscheck.npz — known answers from INDEPENDENT implementations.

The reference's arithmetic lives in OpenCV 4.5.2, which is not in this image, so
the CPU restatement under oracle/ cannot be pinned against OpenCV itself ("parity unpinned").  What
the build container does hold is scikit-image 0.18.3 / SciPy 1.7.1 in /opt/conda (python 3.9): a
different code base implementing some of the same published primitives.  This script records their
outputs on small seeded inputs; tests/test_oracle_crosscheck.py holds the restatement to them.
Only primitives whose definition is implementation-independent are used:

  fast9     segment test FAST-9/16, threshold 20 (Rosten & Drummond): the SET of pixels passing the
            test.  skimage.feature.corner_fast works on floats in [0,1]; threshold (20+0.5)/255 makes
            its strict comparisons equal to the integer test  p > v+20 / p < v-20  for every u8 pair.
  gauss     7-tap sigma-2 separable Gaussian with mirror (reflect-101) border, float64 (scipy.ndimage);
            the restatement's fixed-point result must round to within 1 grey level of it.
  simfit    least-squares similarity transform (Umeyama, skimage SimilarityTransform.estimate) of
            exact + noisy correspondences: the optimum of the objective the LM refine minimises.
  hamming   all-pairs Hamming distances (scipy cdist over unpacked bits) and their stable arg-sort.
  linear    bilinear resize with half-pixel centres and edge clamp (skimage.transform.resize, order 1), float64:
            INTER_LINEAR_EXACT is the same interpolation in fixed point -> within one grey level.
  area      box-filter downscale by integer factors (skimage downscale_local_mean) = INTER_AREA
            when the factor is an integer.

Run with the conda interpreter (the system python has no skimage):
    /opt/conda/bin/python3.9 tests/golden/make_crosscheck.py
"""
import os

import numpy as np
from scipy import ndimage
from scipy.spatial.distance import cdist
from skimage.feature import corner_fast
from skimage.transform import SimilarityTransform, downscale_local_mean

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(0x511DE0)


def textured(h, w):
    """Blocks, glyph-like boxes and noise: plenty of FAST corners of both polarities."""
    img = np.full((h, w), 235, np.int32)
    for _ in range(60):
        y, x = rng.randint(0, h - 8), rng.randint(0, w - 8)
        hh, ww = rng.randint(3, 24), rng.randint(3, 40)
        img[y:y + hh, x:x + ww] = rng.randint(0, 256)
    img += np.rint(rng.normal(0, 6, (h, w))).astype(np.int32)
    yy, xx = np.mgrid[0:h, 0:w]
    img += ((xx * 3 + yy * 2) % 17) - 8            # a ramp so that near-threshold pairs occur
    return np.clip(img, 0, 255).astype(np.uint8)


out = {}

# --- FAST-9 ---------------------------------------------------------------------------------------
g = textured(120, 168)
resp = corner_fast(g, n=9, threshold=20.5 / 255.0)
out["fast_img"] = g
out["fast_mask"] = (resp > 0).astype(np.uint8)

# --- Gaussian 7x7 sigma 2 ---------------------------------------------------------------------------
x = np.arange(7) - 3
k = np.exp(-(x * x) / (2.0 * 2.0 * 2.0))
k /= k.sum()
gi = textured(96, 128)
gf = ndimage.correlate1d(ndimage.correlate1d(gi.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
out["gauss_img"] = gi
out["gauss_f64"] = gf

# --- similarity fit ---------------------------------------------------------------------------------
n = 200
src = rng.uniform(0, 2000, (n, 2))
s, th, tx, ty = 0.93, np.deg2rad(0.7), 41.5, -12.25
R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
dst_exact = s * src @ R.T + np.array([tx, ty])
dst_noisy = dst_exact + rng.normal(0, 0.4, (n, 2))
src32, dste32, dstn32 = src.astype(np.float32), dst_exact.astype(np.float32), dst_noisy.astype(np.float32)
for name, d in (("exact", dste32), ("noisy", dstn32)):
    t = SimilarityTransform()
    assert t.estimate(src32.astype(np.float64), d.astype(np.float64))
    out["simfit_" + name] = t.params[:2, :].copy()
out["simfit_src"] = src32
out["simfit_dst_exact"] = dste32
out["simfit_dst_noisy"] = dstn32

# --- Hamming ------------------------------------------------------------------------------------------
q = rng.randint(0, 256, (37, 32)).astype(np.uint8)
t = rng.randint(0, 256, (211, 32)).astype(np.uint8)
t[5] = q[3]; t[100] = q[3]; t[17] = q[0] ^ np.uint8(1)          # zero distances, ties
D = np.rint(cdist(np.unpackbits(q, axis=1), np.unpackbits(t, axis=1), "hamming") * 256).astype(np.int32)
order = np.argsort(D, axis=1, kind="stable")[:, :30]
out["ham_q"], out["ham_t"] = q, t
out["ham_idx"] = order.astype(np.int32)
out["ham_dist"] = np.take_along_axis(D, order, axis=1)

# --- INTER_AREA at integer factors -----------------------------------------------------------------
a = rng.randint(0, 256, (60, 84, 3)).astype(np.uint8)
out["area_img"] = a
out["area_2"] = downscale_local_mean(a.astype(np.float64), (2, 2, 1))
out["area_3"] = downscale_local_mean(a.astype(np.float64), (3, 3, 1))

# --- bilinear resize, half-pixel centres, edge clamp (one pyramid step, factor 1.2) -------------------
from skimage.transform import resize  # noqa: E402
li = textured(90, 126)
out["lin_img"] = li
out["lin_f64"] = resize(li.astype(np.float64), (75, 105), order=1, mode="edge", anti_aliasing=False, preserve_range=True)

np.savez_compressed(os.path.join(HERE, "crosscheck.npz"), **out)
print("fast corners:", int(out["fast_mask"].sum()), "of", g.size)
print({k: v.shape for k, v in out.items()})
