"""The oracle against every constant SURVEY.md Appendix B derives from the
OpenCV 4.5.2 recurrences (the only pins that exist: the reference holds no
golden vector for this path — parity with OpenCV itself is unpinned)."""
import ctypes as C
import hashlib

import numpy as np


def test_config_defaults_are_reference_literals(oracle):
    c = oracle.default_config()
    # crates/matching-opencv/src/feature_extractor.rs:14-22
    assert (c.nfeatures, c.nlevels, c.edge_threshold, c.patch_size, c.fast_threshold) == (2000, 8, 62, 62, 20)
    assert abs(c.scale_factor - 1.2) < 1e-7
    # lib.rs:266,275,295,330,333,381 ; image_utils.rs:11,52 ; video_capture.rs:98
    assert c.knn_k == 30 and abs(c.vote_tolerance - 1.05) < 1e-7 and c.max_candidate_pages == 40
    assert (c.ransac_threshold, c.ransac_max_iters, c.ransac_confidence, c.refine_iters) == (3.0, 2000, 0.99, 10)
    assert (c.max_rated, c.min_rating, c.min_rating_ratio) == (10, 50.0, 0.2)
    assert abs(c.min_similarity - 0.5) < 1e-7 and c.small_area == 120000
    assert abs(c.changed_similarity - 0.98) < 1e-7


def test_level_quotas_B1(oracle):
    for nf, want in [(500, [109, 90, 75, 63, 52, 44, 36, 31]), (1000, [217, 181, 151, 126, 105, 87, 73, 60]),
                     (2000, [434, 362, 302, 251, 209, 175, 145, 122])]:
        assert list(oracle.level_quotas(oracle.default_config(nfeatures=nf))) == want


def test_pyramid_sizes_B1(oracle):
    c = oracle.default_config()
    table = {
        (640, 360): ([640, 533, 444, 370, 309, 257, 214, 179], [360, 300, 250, 208, 174, 145, 121, 100], 713085),
        (1920, 1080): ([1920, 1600, 1333, 1111, 926, 772, 643, 536], [1080, 900, 750, 625, 521, 434, 362, 301], 6419321),
        (2001, 1125): ([2001, 1667, 1390, 1158, 965, 804, 670, 558], [1125, 937, 781, 651, 543, 452, 377, 314], 6967757),
        (3840, 2160): ([3840, 3200, 2667, 2222, 1852, 1543, 1286, 1072], [2160, 1800, 1500, 1250, 1042, 868, 723, 603], 25677702),
    }
    for (w, h), (ws, hs, total) in table.items():
        a, b, _ = oracle.pyramid_sizes(w, h, c)
        assert list(a) == ws and list(b) == hs
        assert int((a.astype(np.int64) * b).sum()) == total


def test_umax_B3(oracle):
    want = [31, 31, 31, 31, 31, 31, 30, 30, 30, 30, 29, 29, 29, 28, 28, 27, 27, 26, 25, 24, 24, 23, 22, 21, 20, 18,
            17, 16, 14, 12, 9, 5]
    assert list(oracle.umax(31))[:32] == want


def test_brief_pattern_B2(oracle):
    p = oracle.brief_pattern(62)
    assert list(p[:16]) == [21, 14, -13, -26, 11, 16, -17, 13, 4, 3, -16, -21, -8, -1, -25, -18]
    assert p.min() >= -31 and p.max() <= 31
    assert hashlib.sha256(p.astype("<i4").tobytes()).hexdigest() == \
        "9a87803116f99fdd4778b096548468c7704e2d112d78bab08079f548d0bbc9a1"


def test_gauss_kernel_A6(oracle):
    assert list(oracle.gauss_kernel(7, 2.0)) == [18, 34, 48, 56, 48, 34, 18]


def test_ransac_rng_draws_A9(oracle):
    st = C.c_uint64(2 ** 64 - 1)
    assert [oracle.lib().so_rng_uniform(C.byref(st), 0, 100) for _ in range(6)] == [5, 4, 40, 73, 31, 12]


def test_small_sizes_B4(oracle):
    for wh in [(1920, 1080), (2001, 1125), (640, 360), (3840, 2160)]:
        assert oracle.small_size(*wh) == (461, 259)
