#!/usr/bin/env python3
"""Dump what REAL OpenCV computes for every primitive of the hot path, so that the CPU restatement (oracle/) — and with it
the HIP kernels, which are bit-exact against it — can be pinned against the library the reference actually runs.

The reference's arithmetic lives in OpenCV 4.5.2 C++ (crate `opencv` 0.52, Cargo.lock:1723-1724; .github/workflows/ci.yml:18),
which exists neither under the reference repository nor in the build image (no cv2, no headers, no network).  Run this
wherever `import cv2` works — ideally `cv2.__version__ == "4.5.2"` built like ci/install-bionic.sh (SSE3 baseline, AVX2
dispatch, IPP on) — from the repository root:

    python tools/pin_opencv.py [--out DIR]     # writes tests/golden/opencv/{meta.json, <image>.npz, points.npz}

and commit the directory.  tests/test_opencv_pin.py then (a) names, per slideo_ocv_variants switch (include/slideo_amd.h),
the value whose restatement reproduces OpenCV bit for bit, (b) fails if that value is not the default, and (c) compares the
end-to-end ORB keypoints / descriptors and estimateAffinePartial2D results.  Without the directory that test SKIPS with a
message saying so (parity stays "unpinned").

Self-contained on purpose (numpy + cv2 + PIL only; nothing of this repository is imported), so that the dump cannot be
contaminated by the code it is meant to check.  Inputs: the reference's own five fixture PNGs (data/matchings/test1,
byte-identical copies in tests/golden/).  The calls mirror the reference's call sites:
  feature_extractor.rs:13-23,32-40  ORB::create(2000, 1.2, 8, 62, 0, 2, FAST_SCORE, 62, 20).detectAndCompute
  image_utils.rs:17                 resize(INTER_AREA) to the to_small_image size
  image_utils.rs:23                 norm(a, b, NORM_L2)
  image_utils.rs:52                 estimateAffinePartial2D(from, to, RANSAC, 3.0, 2000, 0.99, 10)
  lib.rs:339-347                    warpAffine(frame, M, slide.size, WARP_INVERSE_MAP, BORDER_CONSTANT, 0)
plus the building blocks ORB uses internally (cvtColor, resize INTER_LINEAR_EXACT, FAST, GaussianBlur / sepFilter2D,
fastAtan2), called the way orb.cpp calls them as far as the Python binding allows.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
OUT = os.path.join(GOLD, "opencv")
IMAGES = ["1-frame.png", "1-slide.png", "2-frame.png", "3-frame.png", "3-slide.png"]
NLEVELS, SCALE = 8, 1.2


def level_sizes(w, h):
    out = []
    for l in range(NLEVELS):
        s = np.float32(np.float64(np.float32(SCALE)) ** l)
        out.append((int(np.rint(np.float32(w) / s)), int(np.rint(np.float32(h) / s))))      # cvRound: ties to even
    return out


def lcg_points(seed, n, w, h, a, b, tx, ty, outlier_frac):
    """seeded point pairs (from -> to = similarity + noise, some outliers) with a plain LCG: no dependence on numpy's RNG streams"""
    st = seed & 0xFFFFFFFF
    def nxt():
        nonlocal st
        st = (st * 1664525 + 1013904223) & 0xFFFFFFFF
        return st / 4294967296.0
    frm = np.zeros((n, 2), np.float32); to = np.zeros((n, 2), np.float32)
    for i in range(n):
        x, y = nxt() * w, nxt() * h
        frm[i] = (x, y)
        if nxt() < outlier_frac:
            to[i] = (nxt() * w, nxt() * h)
        else:
            to[i] = (a * x - b * y + tx + (nxt() - 0.5) * 2.0, b * x + a * y + ty + (nxt() - 0.5) * 2.0)
    return frm, to


def main():
    import cv2
    from PIL import Image
    global OUT
    if len(sys.argv) > 2 and sys.argv[1] == "--out":          # another target directory (tests/test_pin_harness_selfcheck.py)
        OUT = sys.argv[2]
    os.makedirs(OUT, exist_ok=True)
    meta = {"cv2_version": cv2.__version__, "build_information": cv2.getBuildInformation(),
            "numpy": np.__version__, "images": IMAGES,
            "note": "produced by tools/pin_opencv.py; the parity pin of oracle/ (SURVEY.md section 8c)"}
    if cv2.__version__ != "4.5.2":
        print("WARNING: cv2 %s, the reference pins 4.5.2 — the dump is still useful but says so in meta.json" % cv2.__version__, file=sys.stderr)
    orb = cv2.ORB_create(2000, 1.2, 8, 62, 0, 2, cv2.ORB_FAST_SCORE, 62, 20)
    fast = cv2.FastFeatureDetector_create(20, True, cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
    k32 = cv2.getGaussianKernel(7, 2, cv2.CV_32F)
    for name in IMAGES:
        bgr = np.ascontiguousarray(np.array(Image.open(os.path.join(GOLD, name)).convert("RGB"))[:, :, ::-1])
        h, w = bgr.shape[:2]
        d = {}
        gray = cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)
        d["gray"] = gray
        prev = gray
        for l, (lw, lh) in enumerate(level_sizes(w, h)):
            if l > 0:
                prev = cv2.resize(prev, (lw, lh), interpolation=cv2.INTER_LINEAR_EXACT)      # orb.cpp: progressive, from the previous level
            if l in (0, 1, 4, 7):
                d["level%d" % l] = prev
                # the blur ORB applies to a level.  orb.cpp blurs a SUBMATRIX of its pyramid buffer; the Python binding cannot
                # make a Mat with SUBMATRIX_FLAG, so three forms are dumped and the ORB descriptors below arbitrate:
                d["blur_gaussianblur_level%d" % l] = cv2.GaussianBlur(prev, (7, 7), 2, sigmaY=2, borderType=cv2.BORDER_REFLECT_101)
                d["blur_sepfilter_f32kernel_level%d" % l] = cv2.sepFilter2D(prev, cv2.CV_8U, k32, k32, borderType=cv2.BORDER_REFLECT_101)
                kps = fast.detect(prev, None)
                d["fast_xy_score_level%d" % l] = np.array([[kp.pt[0], kp.pt[1], kp.response] for kp in kps], np.float32).reshape(-1, 3)
        kps, desc = orb.detectAndCompute(bgr, None)
        d["orb_kp"] = np.array([[kp.pt[0], kp.pt[1], kp.size, kp.angle, kp.response, kp.octave] for kp in kps], np.float32).reshape(-1, 6)
        d["orb_desc"] = desc if desc is not None else np.zeros((0, 32), np.uint8)
        factor = np.sqrt(np.float32(120000.0) / np.float32(w * h), dtype=np.float32)
        sw, sh = int(np.float32(w) * factor), int(np.float32(h) * factor)                    # image_utils.rs:11-16
        d["small"] = cv2.resize(bgr, (sw, sh), interpolation=cv2.INTER_AREA)
        # re-projection: warp into a 2001 x 1125 slide space with fixed transforms, then small image, then L2 norm
        for j, M in enumerate([[0.9593, -0.0008, 0.92, 0.0008, 0.9593, -0.40], [1.04, 0.013, -31.7, -0.013, 1.04, 12.3]]):
            Mm = np.array(M, np.float64).reshape(2, 3)
            warped = cv2.warpAffine(bgr, Mm, (2001, 1125), flags=cv2.WARP_INVERSE_MAP, borderMode=cv2.BORDER_CONSTANT, borderValue=0)
            d["warp%d_M" % j] = Mm
            d["warp%d_crop" % j] = warped[300:420, 800:1000].copy()
            ws = cv2.resize(warped, (461, 259), interpolation=cv2.INTER_AREA)
            d["warp%d_small" % j] = ws
            d["warp%d_norm_vs_small" % j] = np.float64(cv2.norm(ws, cv2.resize(bgr, (461, 259), interpolation=cv2.INTER_AREA), cv2.NORM_L2))
        np.savez_compressed(os.path.join(OUT, name.replace(".png", ".npz")), **d)
        print(name, "keypoints", len(kps))
    # fastAtan2 and estimateAffinePartial2D on seeded inputs
    pts = {}
    st = 12345
    ys, xs = [], []
    for i in range(4000):
        st = (st * 1664525 + 1013904223) & 0xFFFFFFFF; y = (st / 4294967296.0 - 0.5) * 2e6
        st = (st * 1664525 + 1013904223) & 0xFFFFFFFF; x = (st / 4294967296.0 - 0.5) * 2e6
        ys.append(np.float32(round(y))); xs.append(np.float32(round(x)))
    pts["atan_y"] = np.array(ys, np.float32); pts["atan_x"] = np.array(xs, np.float32)
    pts["atan_deg"] = np.array([cv2.fastAtan2(float(y), float(x)) for y, x in zip(ys, xs)], np.float32)
    cases = [(1, 400, 0.0), (2, 400, 0.3), (3, 120, 0.6), (4, 2, 0.0), (5, 3, 0.0), (6, 1500, 0.85), (7, 60, 0.5)]
    for seed, n, of in cases:
        frm, to = lcg_points(seed, n, 2001, 1125, 0.95, 0.01, 3.0, -2.0, of)
        M, inl = cv2.estimateAffinePartial2D(frm, to, method=cv2.RANSAC, ransacReprojThreshold=3.0, maxIters=2000, confidence=0.99, refineIters=10)
        pts["aff%d_from" % seed] = frm; pts["aff%d_to" % seed] = to
        pts["aff%d_M" % seed] = np.zeros((2, 3)) if M is None else M
        pts["aff%d_inliers" % seed] = np.zeros(n, np.uint8) if inl is None else inl.reshape(-1).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "points.npz"), **pts)
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
