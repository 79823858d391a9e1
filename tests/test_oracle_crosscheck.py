"""The CPU restatement held to INDEPENDENT implementations (scikit-image 0.18.3 / SciPy 1.7.1).

OpenCV — where the reference's arithmetic lives — is not available, so parity with it stays unpinned
(DESIGN.md §5).  These known answers come from a different code base implementing the same published
primitives; tests/golden/make_crosscheck.py (run under /opt/conda's python 3.9) wrote them.  They pin
the definition of each primitive, not OpenCV's rounding choices.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as o  # noqa: E402


@pytest.fixture(scope="module")
def xc():
    return np.load(os.path.join(ROOT, "tests", "golden", "crosscheck.npz"))


def test_fast9_corner_set_equals_skimage(xc):
    """The segment test (9 contiguous of 16, strict, threshold 20): same pixel set, both polarities."""
    img, want = xc["fast_img"], xc["fast_mask"].astype(bool)
    got = o.fast_score_map(img, 20) > 0
    assert want[3:-3, 3:-3].sum() > 500
    assert np.array_equal(got[3:-3, 3:-3], want[3:-3, 3:-3])
    assert not got[:3].any() and not got[-3:].any() and not got[:, :3].any() and not got[:, -3:].any()
    # corners score at least the threshold (cornerScore = max(t, arcs) - 1), never below
    assert (o.fast_score_map(img, 20)[got] >= 19).all()


def test_gaussian7_within_one_level_of_float(xc):
    """Fixed-point 7x7 sigma-2 blur with reflect-101 border vs the float64 separable filter."""
    img, ref = xc["gauss_img"], xc["gauss_f64"]
    got = o.gaussian_blur7(img).astype(np.float64)
    err = np.abs(got - ref)
    assert err.max() <= 1.5, err.max()          # 8-bit kernel weights: a few pixels land > 1 level off the ideal
    assert err.mean() < 0.3 and (err > 1.0).mean() < 1e-3
    # against the same separable filter run in float64 with the restatement's own quantised weights the only
    # difference left is rounding: at most one level
    from scipy import ndimage
    kq = o.gauss_kernel(7, 2.0).astype(np.float64) / 256.0
    rq = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), kq, axis=1, mode="mirror"), kq, axis=0, mode="mirror")
    assert np.abs(got - rq).max() <= 1.0
    k = o.gauss_kernel(7, 2.0).astype(np.float64)
    x = np.arange(7) - 3
    kf = np.exp(-(x * x) / 8.0); kf /= kf.sum()
    assert k.sum() == 256 and np.abs(k / 256.0 - kf).max() < 1.0 / 256      # 8-bit weights, each within one step


@pytest.mark.parametrize("which,tol", [("exact", 2e-4), ("noisy", 2e-3)])
def test_similarity_fit_reaches_umeyama_optimum(xc, which, tol):
    """All correspondences are inliers -> RANSAC's LM refine must land on the least-squares similarity."""
    cfg = o.default_config()
    found, M, mask, _ = o.estimate_affine_partial(xc["simfit_src"], xc["simfit_dst_" + which], cfg)
    want = xc["simfit_" + which]
    assert found and mask.all()
    assert abs(M[0, 0] - M[1, 1]) < 1e-12 and abs(M[0, 1] + M[1, 0]) < 1e-12     # 4-DOF form
    assert np.abs(M[:, :2] - want[:, :2]).max() < tol * 1e-2
    assert np.abs(M[:, 2] - want[:, 2]).max() < tol * 50


def test_hamming_knn_equals_scipy(xc):
    idx, dist = o.knn_hamming(xc["ham_q"], xc["ham_t"], 30)
    assert np.array_equal(dist.astype(np.int32), xc["ham_dist"])
    assert np.array_equal(idx, xc["ham_idx"])                  # ties: lowest train row first


@pytest.mark.parametrize("f", [2, 3])
def test_inter_area_integer_factor_is_block_mean(xc, f):
    img = xc["area_img"]
    got = o.resize_area(img, img.shape[1] // f, img.shape[0] // f).astype(np.float64)
    want = xc["area_%d" % f]
    assert np.abs(got - want).max() <= 0.5 + 1e-9


def test_linear_exact_within_one_level_of_float_bilinear(xc):
    """One pyramid step (1.2x down): same half-pixel-centre bilinear sample positions and edge clamp as skimage's
    float resize; the fixed-point weights and two rounding stages stay within one grey level."""
    img, ref = xc["lin_img"], xc["lin_f64"]
    got = o.resize_linear_exact(img, ref.shape[1], ref.shape[0]).astype(np.float64)
    err = np.abs(got - ref)
    assert err.max() <= 1.0, err.max()
    assert err.mean() < 0.3
