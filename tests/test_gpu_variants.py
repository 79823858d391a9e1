"""HIP path vs the CPU restatement for every value of the OpenCV-variant switches the library implements
(slideo_ocv_variants, include/slideo_amd.h): the switch must select the SAME restatement on both sides, bit for bit."""
import os

import numpy as np
import pytest
from PIL import Image

from conftest import small_cfg

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _natural():
    img = np.array(Image.open(os.path.join(HERE, "golden", "2-frame.png")).convert("RGB"))
    return np.ascontiguousarray(img[100:900, 300:1700, ::-1])          # 1400 x 800, colourful


VARIANTS = [dict(), dict(ocv_blur=1), dict(ocv_blur=2), dict(ocv_blur=3), dict(ocv_gray=1), dict(ocv_resize=1), dict(ocv_atan=1),
            dict(ocv_area=1), dict(ocv_rng_mul=4164903691),
            dict(ocv_blur=3, ocv_gray=1, ocv_resize=1, ocv_atan=1, ocv_area=1), dict(ocv_blur=1, ocv_atan=1)]


@pytest.mark.parametrize("over", VARIANTS, ids=lambda d: ",".join("%s%s" % (k[4:], v) for k, v in d.items()) or "default")
def test_orb_and_blurred_pyramid_bit_exact_per_variant(capi, oracle, cfg0_data, over):
    pages, frames, _, _ = cfg0_data
    m = capi.Matcher(capi.default_config(nfeatures=700, **over))
    ocfg = oracle.default_config(nfeatures=700, **over)
    for img in (_natural(), frames[1], pages[1]):          # (frames[0] is a "no slide" frame: no keypoints)
        for lvl in (0, 3, 7):
            for blurred in (False, True):
                assert np.array_equal(m.pyramid_level(img, lvl, blurred), oracle.pyramid_level(img, ocfg, lvl, blurred)), (lvl, blurred)
        gk, gd = m.orb(img)
        ok, od = oracle.orb(img, ocfg)
        assert len(gk) == len(ok) > 50
        for f in ("x", "y", "size", "angle", "response", "octave"):
            assert np.array_equal(gk[f], ok[f]), f
        assert np.array_equal(gd, od)
        if not over.get("ocv_area"):
            assert np.array_equal(m.small_image(img), oracle.small_image(img))
    m.close()


def test_area_variant_small_image(capi, oracle):
    import ctypes as C
    img = _natural()
    m = capi.Matcher(capi.default_config(ocv_area=1))
    g = m.small_image(img)
    h, w, _ = img.shape
    sw, sh = oracle.small_size(w, h)
    o = np.empty((sh, sw, 3), np.uint8)
    assert oracle.lib().so_resize_area_bgr8_v(img.ctypes.data_as(C.c_void_p), w, h, w * 3, o.ctypes.data_as(C.c_void_p), sw, sh, 1) == 0
    assert np.array_equal(g, o)
    m.close()


@pytest.mark.parametrize("over", [dict(), dict(ocv_blur=2), dict(ocv_blur=3, ocv_gray=1, ocv_resize=1, ocv_atan=1, ocv_area=1)],
                         ids=["default", "blur2", "all-alternatives"])
def test_end_to_end_traces_per_variant(capi, oracle, cfg0_data, over):
    from test_gpu_parity import _build_both, _compare_traces
    pages, frames, truth, _ = cfg0_data
    m, db = _build_both(capi, oracle, small_cfg(capi, **over), small_cfg(oracle, **over), pages)
    assert m.descriptor_count == db.descriptor_count
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v)
    assert list(v["page_idx"]) == list(truth)
    m.close()


def test_default_blur_is_the_f32_path_and_differs_from_the_q8_forms(capi):
    img = _natural()
    descs = {}
    for b in range(4):
        m = capi.Matcher(capi.default_config(nfeatures=700, ocv_blur=b))
        descs[b] = m.orb(img)[1]
        m.close()
    m = capi.Matcher(capi.default_config(nfeatures=700))
    assert np.array_equal(m.orb(img)[1], descs[0])
    m.close()
    assert descs[0].shape == descs[2].shape == descs[3].shape
    bits = lambda a, b: np.unpackbits(a ^ b).mean()
    assert 0 < bits(descs[0], descs[2]) < 0.08 and 0 < bits(descs[2], descs[3]) < 0.08 and bits(descs[0], descs[1]) < 0.005


def test_unimplemented_variant_values_fail_loudly(capi):
    for over in (dict(ocv_warp=1), dict(ocv_lm=1), dict(ocv_blur=4), dict(ocv_gray=-1)):
        with pytest.raises(capi.SlideoError) as e:
            capi.Matcher(capi.default_config(**over))
        assert e.value.code == 5 and "ocv." in str(e.value)
