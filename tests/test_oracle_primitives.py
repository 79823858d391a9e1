"""Hand-computable cases for each primitive of the CPU restatement."""
import numpy as np
import pytest


def test_gray_coefficients(oracle):
    img = np.zeros((1, 4, 3), np.uint8)
    img[0, 0] = (255, 0, 0); img[0, 1] = (0, 255, 0); img[0, 2] = (0, 0, 255); img[0, 3] = (255, 255, 255)
    g = oracle.gray(img)[0]
    assert list(g) == [(255 * 3735 + 16384) >> 15, (255 * 19235 + 16384) >> 15, (255 * 9798 + 16384) >> 15, 255]


def test_resize_linear_exact_constant_and_identity(oracle):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear_exact(a, 53, 37), a)
    c = np.full((40, 60), 77, np.uint8)
    assert np.all(oracle.resize_linear_exact(c, 50, 33) == 77)


def test_resize_linear_exact_half(oracle):
    # exact 2:1 shrink samples at half-pixel centres: mean of 2x2 with 8.8 weights 128/128
    a = np.arange(64, dtype=np.uint8).reshape(8, 8) * 3
    r = oracle.resize_linear_exact(a, 4, 4)
    want = (a[0::2, 0::2].astype(int) + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2]) * 128 * 128
    assert np.array_equal(r, ((want + 32768) >> 16).astype(np.uint8))


def test_fast_score_on_synthetic_corner(oracle):
    img = np.full((32, 32), 200, np.uint8)
    img[16:, 16:] = 50            # dark quadrant: corner at (16,16)
    sc = oracle.fast_score_map(img, 20)
    assert sc[0:3].sum() == 0 and sc[:, 0:3].sum() == 0          # unscanned border
    # the dark pixel at the tip sees 11 contiguous brighter circle pixels: |diff| = 150 -> score 149
    assert sc[16, 16] == 149
    flat = np.full((32, 32), 90, np.uint8)
    assert oracle.fast_score_map(flat, 20).sum() == 0


def test_fast_nms_strict(oracle):
    # a perfect step corner gives tied scores at neighbouring pixels (none survives strict NMS);
    # textured input gives isolated maxima
    rng = np.random.default_rng(11)
    img = (rng.integers(0, 2, (12, 12)) * 150 + 50).astype(np.uint8).repeat(5, 0).repeat(5, 1)
    img = (img.astype(int) + rng.integers(-9, 10, img.shape)).clip(0, 255).astype(np.uint8)
    nms = oracle.fast_nms_map(img, 20)
    sc = oracle.fast_score_map(img, 20)
    ys, xs = np.nonzero(nms)
    assert len(ys) >= 5
    for y, x in zip(ys, xs):
        nb = sc[y - 1:y + 2, x - 1:x + 2].astype(int).copy()
        c = nb[1, 1]; nb[1, 1] = -1
        assert c > nb.max()


def test_blur_preserves_constant_and_sums(oracle):
    c = np.full((30, 41), 123, np.uint8)
    assert np.all(oracle.gaussian_blur7(c) == 123)
    imp = np.zeros((21, 21), np.uint8); imp[10, 10] = 255
    b = oracle.gaussian_blur7(imp).astype(int)
    k = np.array([18, 34, 48, 56, 48, 34, 18])
    want = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(b[7:14, 7:14], want)


def test_fast_atan2_quadrants(oracle):
    assert oracle.fast_atan2(0.0, 1.0) == 0.0
    assert abs(oracle.fast_atan2(1.0, 0.0) - 90.0) < 1e-4
    assert abs(oracle.fast_atan2(0.0, -1.0) - 180.0) < 1e-4
    assert abs(oracle.fast_atan2(-1.0, 0.0) - 270.0) < 1e-4
    assert abs(oracle.fast_atan2(1.0, 1.0) - 45.0) < 0.02
    assert oracle.fast_atan2(0.0, 0.0) == 0.0


def test_knn_ties_go_to_lower_row(oracle):
    t = np.zeros((6, 32), np.uint8)
    t[1, 0] = 1; t[2, 0] = 1; t[3, 0] = 3; t[4, 0] = 1      # rows 1,2,4 tie at distance 1
    q = np.zeros((1, 32), np.uint8)
    idx, dist = oracle.knn_hamming(q, t, 4)
    assert list(idx[0]) == [0, 5, 1, 2] and list(dist[0]) == [0, 0, 1, 1]
    idx, dist = oracle.knn_hamming(q, t[:2], 4)              # fewer than k train rows
    assert list(idx[0]) == [0, 1, -1, -1] and list(dist[0]) == [0, 1, 65535, 65535]


def test_knn_matches_numpy_sort(oracle):
    rng = np.random.default_rng(7)
    q = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    idx, dist = oracle.knn_hamming(q, t, 30)
    d = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
    order = np.lexsort((np.broadcast_to(np.arange(500), d.shape), d), axis=1)[:, :30]
    assert np.array_equal(idx, order)
    assert np.array_equal(dist, np.take_along_axis(d, order, 1))


def _similarity_pts(rng, n, scale=0.9, ang=0.01, tx=12.0, ty=-7.0, noise=0.3):
    src = rng.uniform(50, 1900, (n, 2)).astype(np.float32)
    c, s = np.cos(ang) * scale, np.sin(ang) * scale
    dst = np.stack([c * src[:, 0] - s * src[:, 1] + tx, s * src[:, 0] + c * src[:, 1] + ty], 1)
    dst = (dst + rng.normal(0, noise, dst.shape)).astype(np.float32)
    return src, dst, np.array([[c, -s, tx], [s, c, ty]])


def test_ransac_exact_similarity_with_outliers(oracle):
    rng = np.random.default_rng(3)
    src, dst, M = _similarity_pts(rng, 300)
    out = rng.choice(300, 120, replace=False)
    dst[out] = rng.uniform(0, 1900, (120, 2)).astype(np.float32)
    found, Mh, mask, iters = oracle.estimate_affine_partial(src, dst, oracle.default_config())
    assert found
    inl = np.ones(300, bool); inl[out] = False
    assert mask[inl].mean() > 0.97 and mask[~inl].mean() < 0.05
    assert np.allclose(Mh, M, rtol=0, atol=5e-2 * np.array([[1e-2, 1e-2, 10], [1e-2, 1e-2, 10]]))
    assert iters < 2000            # adaptive termination kicked in


def test_ransac_degenerate_counts(oracle):
    c = oracle.default_config()
    found, M, mask, _ = oracle.estimate_affine_partial(np.zeros((1, 2)), np.zeros((1, 2)), c)
    assert not found and mask.sum() == 0
    found, M, mask, _ = oracle.estimate_affine_partial(np.zeros((0, 2)), np.zeros((0, 2)), c)
    assert not found
    src = np.array([[0, 0], [10, 0]], np.float32); dst = np.array([[5, 5], [5, 25]], np.float32)
    found, M, mask, _ = oracle.estimate_affine_partial(src, dst, c)        # exactly 2: direct solve
    assert found and list(mask) == [1, 1]
    assert np.allclose(M, [[0, -2, 5], [2, 0, 5]])


def test_warp_identity_and_shift(oracle):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    assert np.array_equal(oracle.warp_affine_nn(img, [[1, 0, 0], [0, 1, 0]], 60, 40), img)
    sh = oracle.warp_affine_nn(img, [[1, 0, 3], [0, 1, 2]], 60, 40)       # dst(x,y) = src(x+3, y+2)
    assert np.array_equal(sh[:38, :57], img[2:, 3:])
    assert sh[39].sum() == 0 and sh[:, 59].sum() == 0                     # BORDER_CONSTANT 0


def test_area_resize_integer_and_fractional(oracle):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    r2 = oracle.resize_area(img, 30, 20)                                   # 2x2 fast path
    want = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(r2, want.astype(np.uint8))
    c = np.full((45, 80, 3), 91, np.uint8)
    assert np.all(oracle.resize_area(c, 37, 21) == 91)                     # weights sum to 1
    r = oracle.resize_area(img, 23, 17).astype(float)
    assert abs(r.mean() - img.mean()) < 1.0


def test_similarity_formula(oracle):
    a = np.zeros((10, 12, 3), np.uint8); b = np.full((10, 12, 3), 255, np.uint8)
    assert oracle.similarity(a, a) == 1.0
    assert abs(oracle.similarity(a, b)) < 1e-6


def test_timeline_dedup(oracle):
    # lib.rs:229-244: sorted by time, consecutive equal pages collapse; None = -1
    t = [0, 5000, 10000, 15000, 20000, 25000]
    p = [3, 3, -1, -1, 4, 3]
    keep = oracle.timeline_dedup(t, p)
    assert list(keep) == [0, 2, 4, 5]
    keep = oracle.timeline_dedup([10, 0, 5], [1, 1, 2])
    assert list(keep) == [1, 2, 0]


def test_knn_l2_u8_known_answers(oracle):
    """Squared-L2 k-NN restatement (north-star extension, BASELINE configs[2]): hand-checkable cases and a numpy cross-check."""
    t = np.zeros((5, 128), np.uint8)
    t[1, 0] = 3; t[2, :2] = (3, 4); t[3] = 255; t[4, 0] = 3          # rows 1 and 4 are duplicates
    q = np.zeros((2, 128), np.uint8); q[1] = 255
    idx, dist = oracle.knn_l2_u8(q, t, 4)
    assert idx[0].tolist() == [0, 1, 4, 2] and dist[0].tolist() == [0, 9, 9, 25]          # tie 9/9 -> lower row first
    assert idx[1, 0] == 3 and dist[1, 0] == 0 and dist[1, 1] == 128 * 255 * 255 - 2 * 255 * 3 - 2 * 255 * 4 + 9 + 16
    idx, dist = oracle.knn_l2_u8(q, t[:2], 4)                                             # fewer rows than k
    assert idx[0].tolist() == [0, 1, -1, -1] and dist[0, 2] == 0xFFFFFFFF
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (40, 128), dtype=np.uint8); t = rng.integers(0, 256, (300, 128), dtype=np.uint8)
    idx, dist = oracle.knn_l2_u8(q, t, 7)
    d = ((q[:, None, :].astype(np.int64) - t[None, :, :].astype(np.int64)) ** 2).sum(2)
    order = np.lexsort((np.broadcast_to(np.arange(300), d.shape), d), axis=1)[:, :7]
    assert np.array_equal(idx, order) and np.array_equal(dist, np.take_along_axis(d, order, 1))


def test_blocked_knn_equals_plain_knn(oracle):
    """The cache-blocked / vectorised k-NN the frame path and bench.py's cpu_baseline run is the plain restatement, bit for bit:
    random rows, duplicates (distance 0, massive ties -> lowest row wins), fewer rows than k, sizes off the 8-row blocks."""
    rng = np.random.default_rng(11)
    for nq, nt, k in ((37, 1001, 30), (5, 7, 30), (64, 4099, 2), (1, 1, 1), (130, 515, 32)):
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        if nt > 100:
            t[50:90] = q[0]; t[10:20] = q[min(3, nq - 1)]; t[nt - 5:] = q[0]
        i0, d0 = oracle.knn_hamming(q, t, k)
        i1, d1, _ = oracle.knn_hamming_blocked(q, t, k)
        assert np.array_equal(i0, i1) and np.array_equal(d0, d1), (nq, nt, k)
