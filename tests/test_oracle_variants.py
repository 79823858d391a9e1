"""The OpenCV-variant switches (slideo_ocv_variants, include/slideo_amd.h) in the CPU restatement: every value of every
switch is a well-formed restatement of the same primitive (tap tables, closeness to the float64 definition, agreement
between alternatives up to their stated rounding difference), and the defaults are the documented ones."""
import ctypes as C

import numpy as np
import pytest


def _lib(oracle):
    L = oracle.lib()
    L.so_fast_atan2_v.restype = C.c_float
    L.so_fast_atan2_v.argtypes = [C.c_float, C.c_float, C.c_int]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_defaults_are_zero_and_cv_rng_coeff(oracle, capi):
    for mod in (oracle, capi):
        o = mod.default_config().ocv
        assert [o.gray, o.blur, o.resize, o.atan, o.warp, o.area, o.lm] == [0] * 7
        assert o.rng_mul == 4164903690
    assert bytes(oracle.default_config()) == bytes(capi.default_config())
    c = capi.default_config(ocv_blur=2, ocv_gray=1)
    assert c.ocv.blur == 2 and c.ocv.gray == 1
    with pytest.raises(AttributeError):
        capi.default_config(ocv_nothing=1)


def test_gaussian_tap_tables(oracle):
    L = _lib(oracle)
    q = np.zeros(7, np.int32)
    L.so_gauss_taps_q8(3, _p(q))
    assert q.tolist() == [18, 34, 48, 56, 48, 34, 18]         # error-diffused, sums to 256 (GaussianBlur bit-exact path)
    L.so_gauss_taps_q8(2, _p(q))
    assert q.tolist() == [18, 34, 49, 55, 49, 34, 18]         # cvRound(k * 256), sums to 257 (sepFilter2D Q8 before 4.2)
    f = np.zeros(7, np.float32)
    L.so_gauss_taps_f32(_p(f))
    x = np.arange(-3, 4, dtype=np.float64)
    ref = np.exp(-x * x / 8.0); ref /= ref.sum()
    assert np.allclose(f, ref, rtol=0, atol=2e-8) and abs(float(f.astype(np.float64).sum()) - 1.0) < 1e-7
    assert np.array_equal(f, f[::-1])


def _blur(oracle, img, v):
    out = np.empty_like(img)
    _lib(oracle).so_gaussian_blur7_v(_p(img), img.shape[1], img.shape[0], _p(out), v)
    return out


def test_blur_variants_against_float64_definition(oracle):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    img[10:30, 20:60] = 255                                   # a saturated patch: the 257-sum taps must clamp, not wrap
    x = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-x * x / 8.0); k /= k.sum()
    pad = np.pad(img.astype(np.float64), 3, mode="reflect")   # numpy 'reflect' == BORDER_REFLECT_101
    tmp = sum(k[i] * pad[:, i:i + img.shape[1]] for i in range(7))
    ref = sum(k[i] * tmp[i:i + img.shape[0], :] for i in range(7))
    outs = [_blur(oracle, img, v) for v in range(4)]
    for v, o in enumerate(outs):
        d = np.abs(o.astype(np.float64) - ref)
        assert d.max() <= (0.51, 0.51, 2.6, 1.6)[v], (v, d.max())     # f32: correctly rounded; Q8 sum 257: (257/256)^2 brighter, <= 2 levels + rounding; Q8 sum 256
    assert outs[2][15, 40] == 255 and outs[3][15, 40] == 255
    # the two f32 forms differ only where a sum sits on a rounding boundary
    assert (outs[0] != outs[1]).mean() < 1e-3
    # and the forms are really different restatements
    assert (outs[0] != outs[2]).mean() > 0.01 and (outs[2] != outs[3]).mean() > 0.01
    assert np.array_equal(oracle.gaussian_blur7(img), outs[3])       # the legacy tap = GaussianBlur's bit-exact path


def test_gray_resize_atan_variants(oracle):
    L = _lib(oracle)
    rng = np.random.default_rng(6)
    bgr = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    g = []
    for v in (0, 1):
        out = np.empty((40, 50), np.uint8)
        L.so_gray_bgr8_v(_p(bgr), 50, 40, 150, _p(out), v)
        g.append(out)
    ref = 0.114 * bgr[..., 0] + 0.587 * bgr[..., 1] + 0.299 * bgr[..., 2]
    for o in g:
        assert np.abs(o - ref).max() <= 0.51
    assert np.array_equal(g[0], oracle.gray(bgr)) and (g[0] != g[1]).any()
    img = rng.integers(0, 256, (60, 72), dtype=np.uint8)
    r = []
    for v in (0, 1):
        out = np.empty((50, 60), np.uint8)
        L.so_resize_linear_exact_v(_p(img), 72, 60, _p(out), 60, 50, v)
        r.append(out)
    assert np.abs(r[0].astype(int) - r[1].astype(int)).max() <= 1
    assert np.array_equal(r[0], oracle.resize_linear_exact(img, 60, 50))
    ys = rng.normal(0, 1000, 2000).astype(np.float32); xs = rng.normal(0, 1000, 2000).astype(np.float32)
    a0 = np.array([L.so_fast_atan2_v(float(y), float(x), 0) for y, x in zip(ys, xs)])
    a1 = np.array([L.so_fast_atan2_v(float(y), float(x), 1) for y, x in zip(ys, xs)])
    ref = np.degrees(np.arctan2(ys.astype(np.float64), xs.astype(np.float64))) % 360
    for a in (a0, a1):
        d = np.abs(a - ref); d = np.minimum(d, 360 - d)
        assert d.max() < 0.02                                  # the polynomial's own error (~0.01 degrees)
    assert np.abs(a0 - a1).max() < 1e-4 and (a0 != a1).any()


def test_warp_area_lm_variants(oracle):
    L = _lib(oracle)
    rng = np.random.default_rng(7)
    src = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    M = np.array([0.93, -0.02, 3.3, 0.02, 0.93, -1.7])
    w = []
    for v in (0, 1):
        out = np.empty((100, 140, 3), np.uint8)
        L.so_warp_affine_nn_bgr8_v(_p(src), 160, 120, 480, _p(M), _p(out), 140, 100, v)
        w.append(out)
    assert np.array_equal(w[0], oracle.warp_affine_nn(src, M, 140, 100))
    assert (w[0] != w[1]).any(axis=2).mean() < 0.02            # only coordinates within 2^-10 of a rounding boundary move
    a = []
    for v in (0, 1):
        out = np.empty((37, 51, 3), np.uint8)
        assert L.so_resize_area_bgr8_v(_p(src), 160, 120, 480, _p(out), 51, 37, v) == 0
        a.append(out)
    assert np.array_equal(a[0], oracle.resize_area(src, 51, 37))
    assert np.abs(a[0].astype(int) - a[1].astype(int)).max() <= 1
    # LM step solvers on a damped normal matrix of the refinement's shape
    for seed in range(20):
        r2 = np.random.default_rng(100 + seed)
        pts = r2.uniform(0, 2000, (60, 2))
        sq = (pts ** 2).sum(); sx = pts[:, 0].sum(); sy = pts[:, 1].sum(); n = 60.0
        A = np.array([[sq, 0, sx, sy], [0, sq, -sy, sx], [sx, -sy, n, 0], [sy, sx, 0, n]], np.float64)
        A += np.diag(np.diag(A)) * 10.0 ** r2.uniform(-6, 0)
        b = r2.normal(0, 1e3, 4)
        x0 = np.zeros(4); x1 = np.zeros(4)
        assert L.so_solve4(_p(np.ascontiguousarray(A)), _p(b), _p(x0), 0) == 1
        assert L.so_solve4(_p(np.ascontiguousarray(A)), _p(b), _p(x1), 1) == 1
        ref = np.linalg.solve(A, b)
        assert np.allclose(x0, ref, rtol=1e-8, atol=1e-12) and np.allclose(x1, ref, rtol=1e-8, atol=1e-12)


def test_variants_reach_the_pipeline_and_range_checks(oracle, cfg0_data):
    import os
    from PIL import Image
    pages, frames, truth, _ = cfg0_data
    # a natural image (the reference's speaker shot): colourful, so every switch has something to act on
    nat = np.array(Image.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "2-frame.png")).convert("RGB"))
    nat = np.ascontiguousarray(nat[200:840, 500:1460, ::-1])
    base_kp, base_desc = oracle.orb(nat, oracle.default_config(nfeatures=500))
    changed = 0
    for over in (dict(ocv_blur=1), dict(ocv_blur=2), dict(ocv_blur=3), dict(ocv_gray=1), dict(ocv_resize=1), dict(ocv_atan=1),
                 dict(ocv_rng_mul=4164903691)):
        kp, desc = oracle.orb(nat, oracle.default_config(nfeatures=500, **over))
        assert abs(len(kp) - len(base_kp)) <= len(base_kp) // 5
        changed += int(len(kp) != len(base_kp) or not np.array_equal(desc, base_desc) or not np.array_equal(kp["angle"], base_kp["angle"]))
    # blur 2 / 3, atan and the RNG constant change descriptors or angles on any image; blur 1 differs from blur 0 on ~1e-4 of
    # the pixels, the gray and resize roundings on a handful of pixels per image: they may leave this image's features untouched
    assert changed >= 4
    # every variant combination still finds the right pages on the synthetic set
    for over in (dict(ocv_blur=3, ocv_gray=1, ocv_resize=1, ocv_atan=1, ocv_area=1, ocv_warp=1, ocv_lm=1), dict(ocv_blur=2)):
        db = oracle.PageDB(oracle.default_config(nfeatures=500, min_rating=12.0, **over))
        db.add_pages(pages, threads=4)
        assert db.finalize() == 0
        assert db.match_frames(frames, threads=4)["page_idx"].tolist() == truth.tolist()
    for bad in (dict(ocv_blur=4), dict(ocv_gray=2), dict(ocv_lm=-1), dict(ocv_warp=2)):
        assert oracle.lib().so_config_supported(C.byref(oracle.default_config(**bad))) == 0


def test_verdict_rule_is_opt_in_and_ranks_by_rating(oracle, synth):
    """slideo_config.verdict_rule (ABI 6): 0 = the reference (best re-projection similarity wins, mo/lib.rs:370-389) is the default;
    1 keeps the survivors' rating order and lets the similarity only accept.  Checked on the restatement's own trace: whatever the
    candidates are, rule 1's verdict is the accepted survivor with the most inliers (first in candidate order among equals) and
    rule 0's the accepted survivor with the highest similarity."""
    assert oracle.default_config().verdict_rule == 0
    pages = synth.pages(6, 800, 450)
    frames, truth, _ = synth.frames(pages, 6, 640, 360)
    seen_difference_possible = False
    for rule in (0, 1):
        kw = dict(nfeatures=500, min_rating=8.0, min_rating_ratio=0.05, verdict_rule=rule)
        db = oracle.PageDB(oracle.default_config(**kw))
        db.add_pages(pages, threads=4)
        assert db.finalize() == 0
        for fr in frames:
            v, c = db.match_frame_trace(fr)
            s = c[c["survived"] == 1]
            ok = s[s["similarity"] > 0.5]
            if len(ok) == 0:
                assert v["page_idx"] == -1
                continue
            seen_difference_possible |= len(ok) > 1
            if rule == 0:
                want = ok[np.argmax(ok["similarity"])]          # (first maximum = stable sort's first)
            else:
                want = ok[np.argmax(ok["inliers"])]             # (candidate order = page order among equal ratings: first maximum)
            assert v["page_idx"] == want["page_idx"] and v["inliers"] == want["inliers"]
    bad = oracle.default_config(verdict_rule=2)
    assert oracle.lib().so_config_supported(C.byref(bad)) == 0
