"""ctypes binding of include/slideo_amd.h (libslideo_amd.so).

The product path.  It fails loudly when the HIP library is missing or no gfx950
device is present; there is no CPU fallback and nothing here imports oracle/.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

OK = 0
ERR_NAMES = {1: "INVALID_ARG", 2: "NO_DEVICE", 3: "HIP", 4: "STATE", 5: "UNSUPPORTED",
             6: "EMPTY_INDEX", 7: "CAPACITY"}


class OcvVariants(C.Structure):
    """slideo_ocv_variants (include/slideo_amd.h): which restatement of each OpenCV primitive runs."""
    _fields_ = [("gray", C.c_int32), ("blur", C.c_int32), ("resize", C.c_int32), ("atan", C.c_int32),
                ("warp", C.c_int32), ("area", C.c_int32), ("lm", C.c_int32), ("rng_mul", C.c_uint32),
                ("hdlt", C.c_int32)]


class Config(C.Structure):
    """slideo_config (include/slideo_amd.h); defaults = the reference's literals."""
    _fields_ = [
        ("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
        ("edge_threshold", C.c_int32), ("patch_size", C.c_int32), ("fast_threshold", C.c_int32),
        ("knn_k", C.c_int32), ("vote_tolerance", C.c_float), ("max_candidate_pages", C.c_int32),
        ("ransac_threshold", C.c_double), ("ransac_max_iters", C.c_int32),
        ("ransac_confidence", C.c_double), ("refine_iters", C.c_int32), ("max_rated", C.c_int32),
        ("min_rating", C.c_double), ("min_rating_ratio", C.c_double), ("min_similarity", C.c_float),
        ("small_area", C.c_int32), ("changed_similarity", C.c_float), ("ratio_test", C.c_float),
        ("verify_model", C.c_int32), ("matcher", C.c_int32), ("lsh_tables", C.c_int32), ("lsh_key_bits", C.c_int32),
        ("lsh_multi_probe", C.c_int32), ("verdict_rule", C.c_int32),
        ("ocv", OcvVariants),
    ]


class SiftConfig(C.Structure):
    """slideo_sift_config (include/slideo_amd.h): cv::SIFT::create's arguments."""
    _fields_ = [("nfeatures", C.c_int32), ("n_octave_layers", C.c_int32), ("contrast_threshold", C.c_double),
                ("edge_threshold", C.c_double), ("sigma", C.c_double)]


def sift_config(**over):
    c = SiftConfig()
    lib().slideo_sift_config_default(C.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    return c


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4")])
VERDICT_DTYPE = np.dtype([("page_idx", "<i4"), ("similarity", "<f4"), ("inliers", "<i4"),
                          ("n_keypoints", "<i4")])
CANDIDATE_DTYPE = np.dtype([("page_idx", "<i4"), ("n_votes", "<i4"), ("inliers", "<i4"),
                            ("survived", "<i4"), ("similarity", "<f4"), ("_pad", "<i4"),
                            ("transform", "<f8", (9,))])

PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p)

EXPORTS = [
    "slideo_abi_version", "slideo_matcher_use_sift", "slideo_matcher_max_in_flight", "slideo_config_default", "slideo_matcher_create", "slideo_matcher_destroy",
    "slideo_last_error", "slideo_matcher_add_pages_bgr8", "slideo_matcher_finalize_pages",
    "slideo_matcher_page_count", "slideo_matcher_descriptor_count", "slideo_matcher_get_page_features",
    "slideo_match_frames_bgr8", "slideo_match_frames_bgr8_dev", "slideo_changed_mask_bgr8",
    "slideo_matcher_set_progress", "slideo_orb_bgr8", "slideo_pyramid_level_bgr8",
    "slideo_knn_hamming", "slideo_knn_l2_u8", "slideo_small_image_bgr8", "slideo_last_frame_candidates",
    "slideo_matcher_set_profiling", "slideo_matcher_read_profile", "slideo_matcher_read_shader_clock", "slideo_matcher_set_knn_engine", "slideo_matcher_set_knn_exact_lists",
    "slideo_match_frames_submit_dev", "slideo_match_frames_collect", "slideo_match_frames_collect_dev",
    "slideo_matcher_add_page_features", "slideo_matcher_get_page_small", "slideo_l2_set_train", "slideo_l2_knn_dev",
    "slideo_matcher_unique_descriptor_count", "slideo_match_kept_frames", "slideo_host_register", "slideo_host_unregister",
    "slideo_sift_config_default", "slideo_sift_bgr8", "slideo_sift_frames_dev", "slideo_sift_layer_bgr8", "slideo_knn_lsh",
    "slideo_device_count", "slideo_device_list", "slideo_group_create", "slideo_group_destroy", "slideo_group_last_error", "slideo_group_device_count",
    "slideo_group_member", "slideo_group_set_progress", "slideo_group_use_sift", "slideo_group_add_pages_bgr8",
    "slideo_group_finalize_pages", "slideo_group_page_count", "slideo_group_descriptor_count", "slideo_group_match_frames_bgr8",
    "slideo_group_last_frame_candidates", "slideo_group_changed_mask_bgr8", "slideo_group_match_kept_frames",
]

_lib = None


class SlideoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("slideo_amd error %d (%s): %s" % (code, ERR_NAMES.get(code, "?"), msg))
        self.code = code


def lib():
    """Loads libslideo_amd.so (building it in-tree if the sources are newer)."""
    global _lib
    if _lib is None:
        path = os.environ.get("SLIDEO_LIB_PATH") or _build.HIP_LIB      # (SLIDEO_LIB_PATH: an experiment build of the same sources)
        if path == _build.HIP_LIB and (not os.path.exists(path) or os.environ.get("SLIDEO_REBUILD")):
            path = _build.build_hip()
        if not os.path.exists(path):
            raise RuntimeError("libslideo_amd.so is missing and could not be built; the HIP "
                               "extension is required (no fallback path exists)")
        L = C.CDLL(path)
        L.slideo_abi_version.restype = C.c_uint32
        L.slideo_last_error.restype = C.c_char_p
        L.slideo_last_error.argtypes = [C.c_void_p]
        L.slideo_matcher_descriptor_count.restype = C.c_int64
        L.slideo_matcher_unique_descriptor_count.restype = C.c_int64
        L.slideo_matcher_descriptor_count.argtypes = [C.c_void_p]
        L.slideo_matcher_page_count.argtypes = [C.c_void_p]
        L.slideo_matcher_max_in_flight.argtypes = [C.c_void_p]
        L.slideo_matcher_destroy.argtypes = [C.c_void_p]
        L.slideo_matcher_destroy.restype = None
        L.slideo_group_last_error.restype = C.c_char_p
        L.slideo_group_last_error.argtypes = [C.c_void_p]
        L.slideo_group_destroy.argtypes = [C.c_void_p]
        L.slideo_group_destroy.restype = None
        L.slideo_group_member.restype = C.c_void_p
        L.slideo_group_member.argtypes = [C.c_void_p, C.c_int32]
        L.slideo_group_device_count.argtypes = [C.c_void_p]
        L.slideo_device_list.argtypes = [C.c_void_p, C.c_int32]
        L.slideo_group_page_count.argtypes = [C.c_void_p]
        L.slideo_group_descriptor_count.argtypes = [C.c_void_p]
        L.slideo_group_descriptor_count.restype = C.c_int64
        _lib = L
    return _lib


def default_config(**over):
    c = Config()
    lib().slideo_config_default(C.byref(c))
    for k, v in over.items():
        tgt, name = (c.ocv, k[4:]) if k.startswith("ocv_") else (c, k)     # ocv_blur=2 -> c.ocv.blur = 2
        if not hasattr(tgt, name):
            raise AttributeError(k)
        setattr(tgt, name, v)
    return c


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _img3(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected an HxWx3 uint8 BGR image")
    return a


class Matcher:
    """Owns one slideo_matcher handle (page database + workspace on one GPU)."""

    def __init__(self, cfg=None, device=0):
        self.cfg = cfg if cfg is not None else default_config()
        self._h = C.c_void_p()
        self._cb = None
        rc = lib().slideo_matcher_create(C.byref(self.cfg), int(device), C.byref(self._h))
        if rc != OK:
            raise SlideoError(rc, lib().slideo_last_error(None).decode())

    def _check(self, rc):
        if rc != OK:
            raise SlideoError(rc, lib().slideo_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().slideo_matcher_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_progress(self, fn):
        """fn(done, total, msg) or None."""
        if fn is None:
            self._cb = None
            self._check(lib().slideo_matcher_set_progress(self._h, None, None))
            return
        self._cb = PROGRESS_FN(lambda user, d, t, msg: fn(int(d), int(t), (msg or b"").decode()))
        self._check(lib().slideo_matcher_set_progress(self._h, self._cb, None))

    # ---- pages -------------------------------------------------------------------
    def add_pages(self, pages):
        pages = [_img3(p) for p in pages]
        n = len(pages)
        ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in pages])
        w = (C.c_int32 * n)(*[p.shape[1] for p in pages])
        h = (C.c_int32 * n)(*[p.shape[0] for p in pages])
        s = (C.c_int32 * n)(*[p.shape[1] * 3 for p in pages])
        self._check(lib().slideo_matcher_add_pages_bgr8(self._h, n, ptrs, w, h, s))

    def page_small(self, page):
        """Small image of a page (to_small_image), as slideo_matcher_get_page_small returns it."""
        sw = C.c_int32(); sh = C.c_int32()
        rc = lib().slideo_matcher_get_page_small(self._h, page, None, C.c_int64(1 << 40), C.byref(sw), C.byref(sh))
        self._check(rc)
        out = np.empty((sh.value, sw.value, 3), np.uint8)
        self._check(lib().slideo_matcher_get_page_small(self._h, page, _p(out), C.c_int64(out.size), C.byref(sw), C.byref(sh)))
        return out

    def add_page_features(self, width, height, kp, desc, small):
        """A page analysed elsewhere (another rank's share of the deck, a cache): slideo_matcher_add_page_features."""
        kp = np.ascontiguousarray(kp, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        small = np.ascontiguousarray(small, np.uint8)
        assert len(kp) == len(desc) and small.ndim == 3 and small.shape[2] == 3
        self._check(lib().slideo_matcher_add_page_features(self._h, int(width), int(height), len(kp), _p(kp), _p(desc), _p(small),
                                                           small.shape[1], small.shape[0]))

    def finalize(self):
        self._check(lib().slideo_matcher_finalize_pages(self._h))

    @property
    def page_count(self):
        return int(lib().slideo_matcher_page_count(self._h))

    @property
    def descriptor_count(self):
        return int(lib().slideo_matcher_descriptor_count(self._h))

    @property
    def unique_descriptor_count(self):
        """Distinct rows among the train descriptors: what the k-NN stage actually searches (results are those of all rows)."""
        return int(lib().slideo_matcher_unique_descriptor_count(self._h))

    def use_sift(self, sift_cfg, ratio=0.0):
        """SIFT features + the squared-L2 search in front of the path's own vote / RANSAC / re-projection stages (north-star /
        configs[2] as a complete matcher).  ratio in (0, 1]: Lowe's ratio test on the two nearest rows; ratio 0: the path's
        tolerance vote on the knn_k nearest rows.  Before the first page."""
        self._check(lib().slideo_matcher_use_sift(self._h, C.byref(sift_cfg), C.c_float(ratio)))
        self._sift = True

    def page_features(self, page):
        n = C.c_int32()
        rc = lib().slideo_matcher_get_page_features(self._h, page, None, None, 0, C.byref(n))
        if rc not in (OK, 7):
            self._check(rc)
        kp = np.zeros(n.value, KEYPOINT_DTYPE)
        desc = np.zeros((n.value, 128 if getattr(self, "_sift", False) else 32), np.uint8)
        self._check(lib().slideo_matcher_get_page_features(self._h, page, _p(kp), _p(desc), n.value, C.byref(n)))
        return kp, desc

    # ---- frames ------------------------------------------------------------------
    def match_frames(self, frames):
        """frames: uint8 [n, h, w, 3] in host memory -> verdict records."""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w, c = frames.shape
        assert c == 3
        out = np.zeros(n, VERDICT_DTYPE)
        self._check(lib().slideo_match_frames_bgr8(self._h, n, _p(frames), w, h, w * 3,
                                                   C.c_int64(w * h * 3), _p(out)))
        return out

    def match_frames_dev(self, dev_ptr, n, w, h, stride=None, frame_stride=None, stream=0):
        """Frames already resident in HBM (raw device pointer)."""
        stride = stride or w * 3
        frame_stride = frame_stride or stride * h
        out = np.zeros(n, VERDICT_DTYPE)
        self._check(lib().slideo_match_frames_bgr8_dev(self._h, n, C.c_void_p(dev_ptr), w, h, stride,
                                                       C.c_int64(frame_stride), _p(out),
                                                       C.c_void_p(stream)))
        return out

    def max_in_flight(self):
        return int(lib().slideo_matcher_max_in_flight(self._h))

    def submit_dev(self, dev_ptr, n, w, h, stride=None, frame_stride=None, stream=0):
        """Streaming form: returns a ticket; at most max_in_flight() units in flight; collect in order."""
        stride = stride or w * 3
        frame_stride = frame_stride or stride * h
        t = C.c_int64()
        self._check(lib().slideo_match_frames_submit_dev(self._h, n, C.c_void_p(dev_ptr), w, h, stride,
                                                         C.c_int64(frame_stride), C.c_void_p(stream), C.byref(t)))
        return (t.value, n)

    def collect(self, ticket, dev_out=0):
        """dev_out: optional device pointer that also receives the n verdict records (16 bytes each)."""
        t, n = ticket
        out = np.zeros(n, VERDICT_DTYPE)
        self._check(lib().slideo_match_frames_collect_dev(self._h, C.c_int64(t), _p(out), C.c_void_p(dev_out or None)))
        return out

    def last_candidates(self, frame_in_batch):
        cands = np.zeros(64, CANDIDATE_DTYPE)
        n = C.c_int32()
        self._check(lib().slideo_last_frame_candidates(self._h, frame_in_batch, _p(cands), 64, C.byref(n)))
        return cands[: n.value].copy()

    def changed_mask(self, frames, prev_small=None):
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w, _ = frames.shape
        changed = np.zeros(n, np.uint8)
        sims = np.zeros(n, np.float32)
        sw, sh = small_size(w, h, self.cfg.small_area)
        last = np.zeros((sh, sw, 3), np.uint8)
        if prev_small is not None:
            prev_small = np.ascontiguousarray(prev_small, np.uint8)
        self._check(lib().slideo_changed_mask_bgr8(self._h, n, _p(frames), w, h, w * 3, C.c_int64(w * h * 3),
                                                   _p(prev_small), _p(last), _p(changed), _p(sims)))
        return changed.astype(bool), sims, last

    # ---- SIFT (north-star extension, csrc/sift.hip.h) ---------------------------------------
    def sift(self, bgr, scfg=None, cap=20000):
        """cv::SIFT::detectAndCompute on one host image: (keypoints, descriptors u8 [n, 128])."""
        bgr = _img3(bgr)
        h, w, _ = bgr.shape
        scfg = scfg or sift_config()
        kp = np.zeros(cap, KEYPOINT_DTYPE); desc = np.zeros((cap, 128), np.uint8)
        n = C.c_int32()
        rc = lib().slideo_sift_bgr8(self._h, C.byref(scfg), _p(bgr), w, h, w * 3, _p(kp), _p(desc), cap, C.byref(n))
        if rc == 7 and n.value > cap:
            return self.sift(bgr, scfg, cap=n.value)
        self._check(rc)
        return kp[: n.value].copy(), desc[: n.value].copy()

    def sift_layer(self, bgr, octave, layer, dog=False, scfg=None):
        bgr = _img3(bgr)
        h, w, _ = bgr.shape
        scfg = scfg or sift_config()
        out = np.empty(4 * h * w, np.float32)
        lw = C.c_int32(); lh = C.c_int32()
        self._check(lib().slideo_sift_layer_bgr8(self._h, C.byref(scfg), _p(bgr), w, h, w * 3, octave, layer, int(dog), _p(out),
                                                 C.c_int64(out.size), C.byref(lw), C.byref(lh)))
        return out[: lw.value * lh.value].reshape(lh.value, lw.value).copy()

    def sift_frames_dev(self, frames_ptr, n, w, h, kp_ptr, desc_ptr, capacity_total, scfg=None):
        """SIFT of n device frames into device arrays; returns (qofs [n + 1] host, kernel ms)."""
        scfg = scfg or sift_config()
        qofs = np.zeros(n + 1, np.uint32)
        ms = C.c_float()
        self._check(lib().slideo_sift_frames_dev(self._h, C.byref(scfg), n, C.c_void_p(frames_ptr), w, h, w * 3, C.c_int64(w * h * 3),
                                                 C.c_int64(capacity_total), C.c_void_p(kp_ptr), C.c_void_p(desc_ptr), _p(qofs), C.byref(ms)))
        return qofs, float(ms.value)

    def match_kept_frames(self, sel):
        """Verdicts of frames `sel` (indices) of the LAST changed_mask call, from the copy that call left on the device."""
        sel = np.ascontiguousarray(sel, np.int32)
        out = np.zeros(len(sel), VERDICT_DTYPE)
        self._check(lib().slideo_match_kept_frames(self._h, len(sel), _p(sel), _p(out)))
        return out

    def set_knn_engine(self, engine):
        """'mfma' (default: FP4 matrix cores, wave shape chosen per launch), 'mfma4' / 'mfma2' (the two shapes forced:
        2 waves/SIMD x 4 query tiles, 4 waves/SIMD x 2 query tiles) or 'valu' (integer popcount); identical results."""
        self._check(lib().slideo_matcher_set_knn_engine(self._h, {"mfma": 0, "valu": 1, "mfma4": 2, "mfma2": 3}[engine]))

    def set_knn_exact_lists(self, on=True):
        """Keep full exact k-NN lists in the matcher (default: only what the 5 % vote can use); same results."""
        self._check(lib().slideo_matcher_set_knn_exact_lists(self._h, int(bool(on))))

    # ---- measurement ------------------------------------------------------------------
    def set_profiling(self, enable=True):
        self._check(lib().slideo_matcher_set_profiling(self._h, int(bool(enable))))

    def read_profile(self):
        """-> dict(stage -> (ms, intervals)), knn_pairs; clears the accumulators."""
        ms = (C.c_double * 4)(); n = (C.c_int64 * 4)(); pairs = C.c_int64()
        self._check(lib().slideo_matcher_read_profile(self._h, ms, n, C.byref(pairs)))
        names = ["orb", "knn", "verify", "total"]
        return {names[i]: (ms[i], n[i]) for i in range(4)}, pairs.value

    def read_shader_clock(self):
        """-> (MHz the search kernel's waves ran at while profiling, blocks that recorded); clears the sums (ABI 7)."""
        mhz = C.c_double(); n = C.c_int64()
        self._check(lib().slideo_matcher_read_shader_clock(self._h, C.byref(mhz), C.byref(n)))
        return mhz.value, n.value

    # ---- debug taps -----------------------------------------------------------------
    def orb(self, bgr, cap=None):
        bgr = _img3(bgr)
        h, w, _ = bgr.shape
        cap = cap or 8192
        while True:
            kp = np.zeros(cap, KEYPOINT_DTYPE)
            desc = np.zeros((cap, 32), np.uint8)
            n = C.c_int32()
            rc = lib().slideo_orb_bgr8(self._h, _p(bgr), w, h, w * 3, _p(kp), _p(desc), cap, C.byref(n))
            if rc == 7 and n.value > cap:                  # SLIDEO_ERR_CAPACITY: *n_out holds the number found
                cap = n.value
                continue
            self._check(rc)
            return kp[: n.value].copy(), desc[: n.value].copy()

    def pyramid_level(self, bgr, level, blurred):
        bgr = _img3(bgr)
        h, w, _ = bgr.shape
        out = np.empty(h * w, np.uint8)
        lw = C.c_int32(); lh = C.c_int32()
        self._check(lib().slideo_pyramid_level_bgr8(self._h, _p(bgr), w, h, w * 3, level, int(blurred), _p(out),
                                                    C.c_int64(out.size), C.byref(lw), C.byref(lh)))
        return out[: lw.value * lh.value].reshape(lh.value, lw.value).copy()

    def knn(self, q, t, k):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.empty((q.shape[0], k), np.int32)
        dist = np.empty((q.shape[0], k), np.uint16)
        self._check(lib().slideo_knn_hamming(self._h, _p(q), q.shape[0], _p(t), t.shape[0], k, _p(idx), _p(dist)))
        return idx, dist

    def l2_set_train(self, t):
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 128)
        self._check(lib().slideo_l2_set_train(self._h, _p(t), t.shape[0]))

    def l2_knn_dev(self, q_dev, nq, k, idx_dev, dist_dev):
        """device pointers in, device pointers out; returns the search kernels' HIP-event time in ms"""
        ms = C.c_float()
        self._check(lib().slideo_l2_knn_dev(self._h, C.c_void_p(q_dev), nq, k, C.c_void_p(idx_dev), C.c_void_p(dist_dev), C.byref(ms)))
        return ms.value

    def knn_lsh(self, q, t, k):
        """The LSH-compatible search (slideo_config.matcher 1) under this matcher's lsh_* parameters."""
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.empty((q.shape[0], k), np.int32)
        dist = np.empty((q.shape[0], k), np.uint16)
        self._check(lib().slideo_knn_lsh(self._h, _p(q), q.shape[0], _p(t), t.shape[0], k, _p(idx), _p(dist)))
        return idx, dist

    def knn_l2_u8(self, q, t, k):
        """Exact squared-L2 k-NN of 128-dim u8 descriptors on the matrix cores (north-star extension, see the header)."""
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 128)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 128)
        idx = np.empty((q.shape[0], k), np.int32)
        dist = np.empty((q.shape[0], k), np.uint32)
        self._check(lib().slideo_knn_l2_u8(self._h, _p(q), q.shape[0], _p(t), t.shape[0], k, _p(idx), _p(dist)))
        return idx, dist

    def small_image(self, bgr):
        bgr = _img3(bgr)
        h, w, _ = bgr.shape
        sw, sh = small_size(w, h, self.cfg.small_area)
        out = np.empty((max(sh, 1), max(sw, 1), 3), np.uint8)
        a = C.c_int32(); b = C.c_int32()
        self._check(lib().slideo_small_image_bgr8(self._h, _p(bgr), w, h, w * 3, _p(out), C.c_int64(out.size),
                                                  C.byref(a), C.byref(b)))
        return out


class Group:
    """slideo_group (include/slideo_amd.h, "N-device group"): one matcher per device behind one handle — page DB replicated,
    a call's pages and frames sharded contiguously over the devices, verdicts gathered into one host array.  Results equal
    a single Matcher's bit for bit.  `devices`: HIP ordinals (may repeat); None or empty = every gfx950 device of the node."""

    def __init__(self, cfg=None, devices=None):
        self.cfg = cfg if cfg is not None else default_config()
        self._h = C.c_void_p()
        self._cb = None
        if devices is None or len(devices) == 0:     # (an empty list = n_devices 0 = the same request: never a stale empty self.devices)
            # every gfx950 device of the node: the library enumerates them by their own HIP ordinals (n_devices 0)
            rc = lib().slideo_group_create(C.byref(self.cfg), 0, None, C.byref(self._h))
            self.devices = device_list() if rc == OK else []
        else:
            self.devices = [int(d) for d in devices]
            arr = (C.c_int32 * len(self.devices))(*self.devices)
            rc = lib().slideo_group_create(C.byref(self.cfg), len(self.devices), arr, C.byref(self._h))
        if rc != OK:
            raise SlideoError(rc, lib().slideo_group_last_error(None).decode())

    def _check(self, rc):
        if rc != OK:
            raise SlideoError(rc, lib().slideo_group_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().slideo_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def member(self, i):
        """Member i as a (non-owning) Matcher: introspection, taps, profiling."""
        m = Matcher.__new__(Matcher)
        m.cfg, m._cb = self.cfg, None
        m._h = C.c_void_p(lib().slideo_group_member(self._h, int(i)))
        m.close = lambda: None                       # the group owns the handle
        if getattr(self, "_sift", False):
            m._sift = True
        return m

    def set_progress(self, fn):
        if fn is None:
            self._cb = None
            self._check(lib().slideo_group_set_progress(self._h, None, None))
            return
        self._cb = PROGRESS_FN(lambda user, d, t, msg: fn(int(d), int(t), (msg or b"").decode()))
        self._check(lib().slideo_group_set_progress(self._h, self._cb, None))

    def use_sift(self, sift_cfg, ratio=0.0):
        self._check(lib().slideo_group_use_sift(self._h, C.byref(sift_cfg), C.c_float(ratio)))
        self._sift = True

    def add_pages(self, pages):
        pages = [_img3(p) for p in pages]
        n = len(pages)
        ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in pages])
        w = (C.c_int32 * n)(*[p.shape[1] for p in pages])
        h = (C.c_int32 * n)(*[p.shape[0] for p in pages])
        s = (C.c_int32 * n)(*[p.shape[1] * 3 for p in pages])
        self._check(lib().slideo_group_add_pages_bgr8(self._h, n, ptrs, w, h, s))

    def finalize(self):
        self._check(lib().slideo_group_finalize_pages(self._h))

    @property
    def page_count(self):
        return int(lib().slideo_group_page_count(self._h))

    @property
    def descriptor_count(self):
        return int(lib().slideo_group_descriptor_count(self._h))

    def match_frames(self, frames):
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w, c = frames.shape
        assert c == 3
        out = np.zeros(n, VERDICT_DTYPE)
        self._check(lib().slideo_group_match_frames_bgr8(self._h, n, _p(frames), w, h, w * 3, C.c_int64(w * h * 3), _p(out)))
        return out

    def last_candidates(self, frame_in_batch):
        cands = np.zeros(64, CANDIDATE_DTYPE)
        n = C.c_int32()
        self._check(lib().slideo_group_last_frame_candidates(self._h, frame_in_batch, _p(cands), 64, C.byref(n)))
        return cands[: n.value].copy()

    def changed_mask(self, frames, prev_small=None):
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w, _ = frames.shape
        changed = np.zeros(n, np.uint8)
        sims = np.zeros(n, np.float32)
        sw, sh = small_size(w, h, self.cfg.small_area)
        last = np.zeros((sh, sw, 3), np.uint8)
        if prev_small is not None:
            prev_small = np.ascontiguousarray(prev_small, np.uint8)
        self._check(lib().slideo_group_changed_mask_bgr8(self._h, n, _p(frames), w, h, w * 3, C.c_int64(w * h * 3),
                                                         _p(prev_small), _p(last), _p(changed), _p(sims)))
        return changed.astype(bool), sims, last

    def match_kept_frames(self, sel):
        sel = np.ascontiguousarray(sel, np.int32)
        out = np.zeros(len(sel), VERDICT_DTYPE)
        self._check(lib().slideo_group_match_kept_frames(self._h, len(sel), _p(sel), _p(out)))
        return out


def device_count():
    """gfx950 devices visible to this process."""
    return int(lib().slideo_device_count())


def device_list():
    """Their HIP ordinals (not necessarily 0 .. count-1 on a node that also holds other architectures)."""
    n = int(lib().slideo_device_list(None, 0))
    arr = (C.c_int32 * max(n, 1))()
    n = min(n, int(lib().slideo_device_list(arr, n)))
    return [int(arr[i]) for i in range(n)]


def small_size(w, h, small_area=120000):
    """to_small_image target size (crates/matching-opencv/src/image_utils.rs:11-16), f32 arithmetic."""
    factor = np.sqrt(np.float32(small_area) / np.float32(w * h), dtype=np.float32)
    return int(np.float32(w) * factor), int(np.float32(h) * factor)
