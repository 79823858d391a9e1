"""ctypes loader for the CPU restatement (oracle/liboracle.so).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module; nothing under slideo_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


class OcvVariants(C.Structure):
    """slideo_ocv_variants (include/slideo_amd.h): which restatement of each OpenCV primitive runs."""
    _fields_ = [("gray", C.c_int32), ("blur", C.c_int32), ("resize", C.c_int32), ("atan", C.c_int32),
                ("warp", C.c_int32), ("area", C.c_int32), ("lm", C.c_int32), ("rng_mul", C.c_uint32),
                ("hdlt", C.c_int32)]


class Config(C.Structure):
    """Mirror of slideo_config (include/slideo_amd.h)."""
    _fields_ = [
        ("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
        ("edge_threshold", C.c_int32), ("patch_size", C.c_int32), ("fast_threshold", C.c_int32),
        ("knn_k", C.c_int32), ("vote_tolerance", C.c_float), ("max_candidate_pages", C.c_int32),
        ("ransac_threshold", C.c_double), ("ransac_max_iters", C.c_int32),
        ("ransac_confidence", C.c_double), ("refine_iters", C.c_int32), ("max_rated", C.c_int32),
        ("min_rating", C.c_double), ("min_rating_ratio", C.c_double), ("min_similarity", C.c_float),
        ("small_area", C.c_int32), ("changed_similarity", C.c_float), ("ratio_test", C.c_float),
        ("verify_model", C.c_int32), ("matcher", C.c_int32), ("lsh_tables", C.c_int32), ("lsh_key_bits", C.c_int32),
        ("lsh_multi_probe", C.c_int32), ("verdict_rule", C.c_int32), ("ocv", OcvVariants),
    ]


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4")])
VERDICT_DTYPE = np.dtype([("page_idx", "<i4"), ("similarity", "<f4"), ("inliers", "<i4"),
                          ("n_keypoints", "<i4")])
CANDIDATE_DTYPE = np.dtype([("page_idx", "<i4"), ("n_votes", "<i4"), ("inliers", "<i4"),
                            ("survived", "<i4"), ("similarity", "<f4"), ("_pad", "<i4"),
                            ("transform", "<f8", (9,))])


class SiftConfig(C.Structure):
    """slideo_sift_config (include/slideo_amd.h): cv::SIFT::create's arguments."""
    _fields_ = [("nfeatures", C.c_int32), ("n_octave_layers", C.c_int32), ("contrast_threshold", C.c_double),
                ("edge_threshold", C.c_double), ("sigma", C.c_double)]


def build(force=False):
    srcs = [os.path.join(_HERE, "slideo_oracle.cpp"), os.path.join(_HERE, "sift_oracle.h")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.so_fast_atan2.restype = C.c_float
        _lib.so_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.so_similarity_bgr8.restype = C.c_float
        _lib.so_pagedb_create.restype = C.c_void_p
        _lib.so_pagedb_descriptor_count.restype = C.c_int64
        _lib.so_rng_next.restype = C.c_uint32
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_config(**over):
    c = Config()
    lib().so_config_default(C.byref(c))
    for k, v in over.items():
        if k.startswith("ocv_"):            # ocv_blur=2 -> c.ocv.blur = 2 (slideo_ocv_variants)
            setattr(c.ocv, k[4:], v)
        else:
            setattr(c, k, v)
    return c


def _img3(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 3
    return a


def gray(bgr):
    bgr = _img3(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().so_gray_bgr8(_p(bgr), w, h, w * 3, _p(out))
    return out


def resize_linear_exact(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((dh, dw), np.uint8)
    lib().so_resize_linear_exact(_p(img), img.shape[1], img.shape[0], _p(out), dw, dh)
    return out


def pyramid_sizes(w, h, cfg):
    ws = np.zeros(cfg.nlevels, np.int32); hs = np.zeros(cfg.nlevels, np.int32)
    sc = np.zeros(cfg.nlevels, np.float32)
    lib().so_pyramid_sizes(w, h, C.byref(cfg), _p(ws), _p(hs), _p(sc))
    return ws, hs, sc


def level_quotas(cfg):
    q = np.zeros(cfg.nlevels, np.int32)
    lib().so_level_quotas(C.byref(cfg), _p(q))
    return q


def umax(half_patch):
    u = np.zeros(half_patch + 2, np.int32)
    lib().so_umax(half_patch, _p(u))
    return u


def brief_pattern(patch_size):
    p = np.zeros(1024, np.int32)
    lib().so_brief_pattern(patch_size, _p(p))
    return p


def gauss_kernel(n, sigma):
    k = np.zeros(n, np.int32)
    lib().so_gauss_kernel(n, C.c_double(sigma), _p(k))
    return k


def fast_score_map(img, thr):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().so_fast_score_map(_p(img), img.shape[1], img.shape[0], thr, _p(out))
    return out


def fast_nms_map(img, thr):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().so_fast_nms_map(_p(img), img.shape[1], img.shape[0], thr, _p(out))
    return out


def gaussian_blur7(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().so_gaussian_blur7(_p(img), img.shape[1], img.shape[0], _p(out))
    return out


def fast_atan2(y, x):
    return float(lib().so_fast_atan2(C.c_float(y), C.c_float(x)))


def pyramid_level(bgr, cfg, level, blurred):
    bgr = _img3(bgr)
    h, w, _ = bgr.shape
    out = np.empty(h * w, np.uint8)
    lw = C.c_int32(); lh = C.c_int32()
    rc = lib().so_pyramid_level(_p(bgr), w, h, w * 3, C.byref(cfg), level, int(blurred), _p(out),
                                C.c_int64(out.size), C.byref(lw), C.byref(lh))
    assert rc == 0, rc
    return out[: lw.value * lh.value].reshape(lh.value, lw.value).copy()


def orb(bgr, cfg, cap=None):
    bgr = _img3(bgr)
    h, w, _ = bgr.shape
    cap = cap or max(8 * cfg.nfeatures, 4096)
    kp = np.zeros(cap, KEYPOINT_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = lib().so_orb_bgr8(_p(bgr), w, h, w * 3, C.byref(cfg), _p(kp), _p(desc), cap)
    assert n >= 0, "unsupported config"
    if n > cap:
        return orb(bgr, cfg, cap=n)
    return kp[:n].copy(), desc[:n].copy()


def sift_config(**over):
    c = SiftConfig()
    lib().so_sift_config_default(C.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    return c


def sift(bgr, scfg=None, cfg=None, cap=None):
    """cv::SIFT::detectAndCompute restated (oracle/sift_oracle.h): (keypoints, descriptors u8 [n,128], stats)."""
    bgr = _img3(bgr)
    h, w, _ = bgr.shape
    scfg = scfg or sift_config()
    cfg = cfg or default_config()
    cap = cap or 20000
    kp = np.zeros(cap, KEYPOINT_DTYPE)
    desc = np.zeros((cap, 128), np.uint8)
    st = np.zeros(3, np.int32)
    n = lib().so_sift_bgr8(_p(bgr), w, h, w * 3, C.byref(scfg), C.byref(cfg.ocv), _p(kp), _p(desc), cap, _p(st))
    if n > cap:
        return sift(bgr, scfg, cfg, cap=n)
    return kp[:n].copy(), desc[:n].copy(), dict(octaves=int(st[0]), extrema=int(st[1]), refined=int(st[2]))


def sift_layer(bgr, octave, layer, dog=False, scfg=None, cfg=None):
    bgr = _img3(bgr)
    h, w, _ = bgr.shape
    scfg = scfg or sift_config()
    cfg = cfg or default_config()
    out = np.empty(4 * h * w, np.float32)
    lw = C.c_int32(); lh = C.c_int32()
    rc = lib().so_sift_layer(_p(bgr), w, h, w * 3, C.byref(scfg), C.byref(cfg.ocv), octave, layer, int(dog), _p(out), C.c_int64(out.size),
                             C.byref(lw), C.byref(lh))
    assert rc == 0, rc
    return out[: lw.value * lh.value].reshape(lh.value, lw.value).copy()


def knn_hamming(q, t, k):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.empty((q.shape[0], k), np.int32)
    dist = np.empty((q.shape[0], k), np.uint16)
    lib().so_knn_hamming(_p(q), q.shape[0], _p(t), t.shape[0], k, _p(idx), _p(dist))
    return idx, dist


def knn_lsh(q, t, k, cfg):
    """slideo_config.matcher 1: the k nearest (distance, row) among each query's LSH candidates; also the tables' bit positions."""
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.empty((q.shape[0], k), np.int32)
    dist = np.empty((q.shape[0], k), np.uint16)
    bits = np.zeros(cfg.lsh_tables * cfg.lsh_key_bits, np.int32)
    lib().so_knn_lsh(_p(q), q.shape[0], _p(t), t.shape[0], k, C.byref(cfg), _p(idx), _p(dist), _p(bits))
    return idx, dist, bits.reshape(cfg.lsh_tables, cfg.lsh_key_bits)


def knn_hamming_blocked(q, t, k):
    """The cache-blocked / AVX-512 VPOPCNTDQ form of knn_hamming (what the frame path runs); returns (idx, dist, simd_used)."""
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.empty((q.shape[0], k), np.int32)
    dist = np.empty((q.shape[0], k), np.uint16)
    simd = lib().so_knn_hamming_blocked(_p(q), q.shape[0], _p(t), t.shape[0], k, _p(idx), _p(dist))
    return idx, dist, bool(simd)


def knn_l2_u8(q, t, k):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 128)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 128)
    idx = np.empty((q.shape[0], k), np.int32)
    dist = np.empty((q.shape[0], k), np.uint32)
    lib().so_knn_l2_u8(_p(q), q.shape[0], _p(t), t.shape[0], k, _p(idx), _p(dist))
    return idx, dist


def estimate_affine_partial(frm, to, cfg):
    frm = np.ascontiguousarray(frm, np.float32).reshape(-1, 2)
    to = np.ascontiguousarray(to, np.float32).reshape(-1, 2)
    n = frm.shape[0]
    M = np.zeros(6, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    it = C.c_int32()
    found = lib().so_estimate_affine_partial(_p(frm), _p(to), n, C.byref(cfg), _p(M), _p(mask),
                                             C.byref(it))
    return bool(found), M.reshape(2, 3), mask[:n].copy(), it.value


def find_homography(frm, to, cfg):
    """cv::findHomography(RANSAC) restated (verify_model 1). Returns (found, H 3x3, mask, stats dict)."""
    frm = np.ascontiguousarray(frm, np.float32).reshape(-1, 2)
    to = np.ascontiguousarray(to, np.float32).reshape(-1, 2)
    n = frm.shape[0]
    H = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    st = np.zeros(4, np.int32)
    found = lib().so_find_homography(_p(frm), _p(to), n, C.byref(cfg), _p(H), _p(mask), _p(st))
    return bool(found), H.reshape(3, 3), mask[:n].copy(), dict(iters=int(st[0]), attempts=int(st[1]), rotations=int(st[2]), draws=int(st[3]))


def homography_dlt(frm, to, variant=0):
    frm = np.ascontiguousarray(frm, np.float32).reshape(-1, 2)
    to = np.ascontiguousarray(to, np.float32).reshape(-1, 2)
    H = np.zeros(9, np.float64)
    n = lib().so_homography_dlt(_p(frm), _p(to), frm.shape[0], _p(H), variant)
    return n, H.reshape(3, 3)


def homography_check_subset(frm, to):
    frm = np.ascontiguousarray(frm, np.float32).reshape(-1, 2)
    to = np.ascontiguousarray(to, np.float32).reshape(-1, 2)
    return bool(lib().so_homography_check_subset(_p(frm), _p(to), frm.shape[0]))


def jacobi_eig(A):
    """cv::eigen on a symmetric matrix (JacobiImpl_): (W descending, V rows = eigenvectors, rotations)."""
    A = np.ascontiguousarray(A, np.float64)
    n = A.shape[0]
    W = np.zeros(n); V = np.zeros((n, n))
    rot = lib().so_jacobi_eig(_p(A), n, _p(W), _p(V))
    return W, V, rot


def warp_perspective_nn(src, H, dw, dh):
    src = _img3(src)
    H = np.ascontiguousarray(H, np.float64).reshape(9)
    out = np.empty((dh, dw, 3), np.uint8)
    lib().so_warp_perspective_nn_bgr8(_p(src), src.shape[1], src.shape[0], src.shape[1] * 3, _p(H), _p(out), dw, dh)
    return out


def warp_affine_nn(src, M, dw, dh):
    src = _img3(src)
    M = np.ascontiguousarray(M, np.float64).reshape(6)
    out = np.empty((dh, dw, 3), np.uint8)
    lib().so_warp_affine_nn_bgr8(_p(src), src.shape[1], src.shape[0], src.shape[1] * 3, _p(M),
                                 _p(out), dw, dh)
    return out


def resize_area(src, dw, dh):
    src = _img3(src)
    out = np.empty((dh, dw, 3), np.uint8)
    rc = lib().so_resize_area_bgr8(_p(src), src.shape[1], src.shape[0], src.shape[1] * 3, _p(out),
                                   dw, dh)
    assert rc == 0, rc
    return out


def small_size(w, h, small_area=120000):
    a = C.c_int32(); b = C.c_int32()
    lib().so_small_size(w, h, small_area, C.byref(a), C.byref(b))
    return a.value, b.value


def small_image(bgr, small_area=120000):
    bgr = _img3(bgr)
    h, w, _ = bgr.shape
    sw, sh = small_size(w, h, small_area)
    out = np.empty((sh, sw, 3), np.uint8)
    a = C.c_int32(); b = C.c_int32()
    rc = lib().so_small_image_bgr8(_p(bgr), w, h, w * 3, small_area, _p(out), C.c_int64(out.size),
                                   C.byref(a), C.byref(b))
    assert rc == 0, rc
    return out


def similarity(a, b):
    a = _img3(a); b = _img3(b)
    assert a.shape == b.shape
    return float(lib().so_similarity_bgr8(_p(a), _p(b), a.shape[1], a.shape[0]))


def changed_mask(frames, cfg, prev_small=None):
    frames = np.ascontiguousarray(frames, np.uint8)
    n, h, w, _ = frames.shape
    sw, sh = small_size(w, h, cfg.small_area)
    changed = np.zeros(n, np.uint8); sims = np.zeros(n, np.float32)
    last = np.zeros((sh, sw, 3), np.uint8)
    pp = _p(np.ascontiguousarray(prev_small, np.uint8)) if prev_small is not None else None
    rc = lib().so_changed_mask_bgr8(_p(frames), n, w, h, w * 3, C.c_int64(w * h * 3), C.byref(cfg),
                                    pp, _p(last), _p(changed), _p(sims))
    assert rc == 0, rc
    return changed.astype(bool), sims, last


class PageDB:
    """Oracle twin of the page side of the matcher (mo/lib.rs:37-64)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self._h = lib().so_pagedb_create(C.byref(cfg))
        assert self._h, "unsupported config"

    def use_sift(self, sift_cfg, ratio=0.75):
        """The product's slideo_matcher_use_sift: SIFT features + squared-L2 2-NN + Lowe's ratio test in front of the path's own
        vote / RANSAC / re-projection stages.  Before the first page."""
        rc = lib().so_pagedb_use_sift(C.c_void_p(self._h), C.byref(sift_cfg), C.c_float(ratio))
        assert rc == 0, rc
        self._sift = True

    def add_page(self, bgr):
        bgr = _img3(bgr)
        h, w, _ = bgr.shape
        rc = lib().so_pagedb_add_page(C.c_void_p(self._h), _p(bgr), w, h, w * 3)
        assert rc == 0, rc

    def add_pages(self, pages, threads=1):
        pages = np.ascontiguousarray(pages, np.uint8)
        n, h, w, _ = pages.shape
        rc = lib().so_pagedb_add_pages(C.c_void_p(self._h), _p(pages), n, w, h, w * 3,
                                       C.c_int64(w * h * 3), threads)
        assert rc == 0, rc

    def finalize(self):
        return lib().so_pagedb_finalize(C.c_void_p(self._h))

    @property
    def descriptor_count(self):
        return int(lib().so_pagedb_descriptor_count(C.c_void_p(self._h)))

    @property
    def page_count(self):
        return int(lib().so_pagedb_page_count(C.c_void_p(self._h)))

    def page_features(self, page):
        n = lib().so_pagedb_get_page_features(C.c_void_p(self._h), page, None, None, 0)
        kp = np.zeros(n, KEYPOINT_DTYPE); desc = np.zeros((n, 128 if getattr(self, "_sift", False) else 32), np.uint8)
        lib().so_pagedb_get_page_features(C.c_void_p(self._h), page, _p(kp), _p(desc), n)
        return kp, desc

    def train(self):
        m = self.descriptor_count
        t = np.zeros((m, 32), np.uint8)
        rc = lib().so_pagedb_get_train(C.c_void_p(self._h), _p(t), C.c_int64(m))
        assert rc == 0
        return t

    def match_frames(self, frames, threads=1):
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w, _ = frames.shape
        out = np.zeros(n, VERDICT_DTYPE)
        rc = lib().so_match_frames(C.c_void_p(self._h), _p(frames), n, w, h, w * 3,
                                   C.c_int64(w * h * 3), _p(out), threads)
        assert rc == 0, rc
        return out

    def match_frame_trace(self, frame):
        frame = _img3(frame)
        h, w, _ = frame.shape
        v = np.zeros(1, VERDICT_DTYPE)
        cands = np.zeros(64, CANDIDATE_DTYPE)
        n = C.c_int32()
        rc = lib().so_match_frame_trace(C.c_void_p(self._h), _p(frame), w, h, w * 3, _p(v),
                                        _p(cands), 64, C.byref(n))
        assert rc == 0, rc
        return v[0], cands[: n.value].copy()

    def __del__(self):
        try:
            lib().so_pagedb_destroy(C.c_void_p(self._h))
        except Exception:
            pass


def timeline_dedup(time_ms, page):
    time_ms = np.ascontiguousarray(time_ms, np.int64)
    page = np.ascontiguousarray(page, np.int32)
    keep = np.zeros(len(page), np.int32)
    m = lib().so_timeline_dedup(_p(time_ms), _p(page), len(page), _p(keep))
    return keep[:m].copy()
