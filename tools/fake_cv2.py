"""A stand-in for `cv2` that implements EXACTLY the calls tools/pin_opencv.py makes, on top of the repository's own CPU
restatement (oracle/pyoracle, default slideo_ocv_variants).

IT PINS NOTHING.  The dump it produces is the restatement talking to itself; its only purpose is to execute the pin harness —
the dumper and the nine checks of tests/test_opencv_pin.py — in an image that has no OpenCV, so that the one route from
"parity unpinned" to "pinned" (run tools/pin_opencv.py where cv2 == 4.5.2, commit tests/golden/opencv/) is known to work
end to end (tests/test_pin_harness_selfcheck.py).  `__version__` says what it is, and the consumer marks any version other than
"4.5.2" as informative only.  Never import this from the product or from a parity test."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pyoracle as _o  # noqa: E402

__version__ = "0.0-restatement-stand-in (NOT OpenCV)"

COLOR_BGR2GRAY = 6
INTER_AREA, INTER_LINEAR_EXACT = 3, 5
BORDER_CONSTANT, BORDER_REFLECT_101 = 0, 4
WARP_INVERSE_MAP = 16
NORM_L2 = 4
RANSAC = 8
CV_8U, CV_32F = 0, 5
ORB_FAST_SCORE = 1
FAST_FEATURE_DETECTOR_TYPE_9_16 = 2


def getBuildInformation():
    return ("General configuration for OpenCV 0.0-restatement-stand-in =====================\n"
            "  tools/fake_cv2.py over oracle/pyoracle (default slideo_ocv_variants): NOT an OpenCV build\n")


class KeyPoint:
    def __init__(self, x, y, size=0.0, angle=-1.0, response=0.0, octave=0):
        self.pt, self.size, self.angle, self.response, self.octave = (float(x), float(y)), float(size), float(angle), float(response), int(octave)


class _Orb:
    def __init__(self, nfeatures, scale, nlevels, edge, first, wta, score, patch, fast_thr):
        assert (first, wta, score) == (0, 2, ORB_FAST_SCORE)
        self.cfg = _o.default_config(nfeatures=nfeatures, scale_factor=scale, nlevels=nlevels, edge_threshold=edge, patch_size=patch,
                                     fast_threshold=fast_thr)

    def detectAndCompute(self, bgr, mask):
        kp, desc = _o.orb(bgr, self.cfg)
        return [KeyPoint(k["x"], k["y"], k["size"], k["angle"], k["response"], k["octave"]) for k in kp], desc


def ORB_create(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, WTA_K, scoreType, patchSize, fastThreshold):
    return _Orb(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, WTA_K, scoreType, patchSize, fastThreshold)


class _Fast:
    def __init__(self, thr, nms, typ):
        assert nms and typ == FAST_FEATURE_DETECTOR_TYPE_9_16
        self.thr = thr

    def detect(self, img, mask):
        m = _o.fast_nms_map(img, self.thr)
        ys, xs = np.nonzero(m)
        return [KeyPoint(x, y, 7.0, -1.0, m[y, x]) for y, x in zip(ys.tolist(), xs.tolist())]


def FastFeatureDetector_create(threshold, nonmaxSuppression, type):
    return _Fast(threshold, nonmaxSuppression, type)


def getGaussianKernel(n, sigma, ktype):
    assert (n, sigma, ktype) == (7, 2, CV_32F)
    x = np.arange(7, dtype=np.float64) - 3.0
    k = np.exp(-0.5 * x * x / 4.0)
    return (k / k.sum()).astype(np.float32).reshape(7, 1)


def cvtColor(bgr, code):
    assert code == COLOR_BGR2GRAY
    return _o.gray(bgr)


def resize(src, dsize, interpolation):
    dw, dh = dsize
    if interpolation == INTER_LINEAR_EXACT:
        return _o.resize_linear_exact(src, dw, dh)
    assert interpolation == INTER_AREA
    return _o.resize_area(src, dw, dh)


def _blur7(img, variant):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    _o.lib().so_gaussian_blur7_v(_o._p(img), img.shape[1], img.shape[0], _o._p(out), variant)
    return out


def GaussianBlur(src, ksize, sigmaX, sigmaY=0, borderType=BORDER_REFLECT_101):
    assert ksize == (7, 7) and sigmaX == 2 and borderType == BORDER_REFLECT_101
    return _blur7(src, 3)                       # a stand-alone Mat takes GaussianBlur's bit-exact fixed-point path (ocv.blur 3)


def sepFilter2D(src, ddepth, kx, ky, borderType=BORDER_REFLECT_101):
    assert ddepth == CV_8U and borderType == BORDER_REFLECT_101
    return _blur7(src, 0)                       # the f32 kernel through sepFilter2D (ocv.blur 0, the default)


def warpAffine(src, M, dsize, flags, borderMode, borderValue):
    assert flags == WARP_INVERSE_MAP and borderMode == BORDER_CONSTANT and borderValue == 0
    return _o.warp_affine_nn(src, np.asarray(M, np.float64).reshape(6), dsize[0], dsize[1])


def norm(a, b, kind):
    assert kind == NORM_L2
    d = a.astype(np.int64) - b.astype(np.int64)
    return float(np.sqrt(float((d * d).sum())))


def fastAtan2(y, x):
    return _o.fast_atan2(y, x)


def estimateAffinePartial2D(frm, to, method, ransacReprojThreshold, maxIters, confidence, refineIters):
    assert method == RANSAC
    cfg = _o.default_config(ransac_threshold=ransacReprojThreshold, ransac_max_iters=maxIters, ransac_confidence=confidence, refine_iters=refineIters)
    found, M, mask, _ = _o.estimate_affine_partial(frm, to, cfg)
    return (M if found else None), (mask.reshape(-1, 1) if found else None)
