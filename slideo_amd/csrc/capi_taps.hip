// capi_taps.hip — debug taps of the parity tests (no kernel of its own).
#include "runtime.hpp"

using namespace slideo;

extern "C" {

// ---- debug taps ---------------------------------------------------------------------------

int32_t slideo_orb_bgr8(slideo_matcher* m, const uint8_t* bgr, int32_t width, int32_t height, int32_t stride_bytes,
                        slideo_keypoint* kp, uint8_t* desc32, int32_t capacity, int32_t* n_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (!bgr || !n_out) fail(SLIDEO_ERR_INVALID_ARG, "null image/n_out");
    validate_image(width, height, stride_bytes);
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    hipStream_t st = S.st;
    const size_t fb = (size_t)height * stride_bytes;
    stage_for_upload(m, fb);
    HIP_CHECK(hipMemcpyAsync(S.d_stage.p, bgr, fb, hipMemcpyHostToDevice, st));
    run_orb(m, S, S.d_stage.as<uint8_t>(), 1, width, height, stride_bytes, (int64_t)fb, false);
    const uint32_t q = S.orb.qtot;
    *n_out = (int32_t)q;
    if ((int64_t)q > capacity) fail(SLIDEO_ERR_CAPACITY, "%u keypoints, capacity %d", q, capacity);
    if (q) {
        if (kp) HIP_CHECK(hipMemcpyAsync(kp, S.d_kp.p, (size_t)q * sizeof(slideo_keypoint), hipMemcpyDeviceToHost, st));
        if (desc32) HIP_CHECK(hipMemcpyAsync(desc32, S.d_desc.p, (size_t)q * 32, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    API_CATCH(m)
}

int32_t slideo_pyramid_level_bgr8(slideo_matcher* m, const uint8_t* bgr, int32_t width, int32_t height, int32_t stride_bytes,
                                  int32_t level, int32_t blurred, uint8_t* out, int64_t out_capacity, int32_t* lw, int32_t* lh) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (!bgr || !out || !lw || !lh) fail(SLIDEO_ERR_INVALID_ARG, "null argument");
    validate_image(width, height, stride_bytes);
    if (level < 0 || level >= m->cfg.nlevels) fail(SLIDEO_ERR_INVALID_ARG, "level out of range");
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    hipStream_t st = S.st;
    const size_t fb = (size_t)height * stride_bytes;
    stage_for_upload(m, fb);
    HIP_CHECK(hipMemcpyAsync(S.d_stage.p, bgr, fb, hipMemcpyHostToDevice, st));
    run_orb(m, S, S.d_stage.as<uint8_t>(), 1, width, height, stride_bytes, (int64_t)fb, false, blurred != 0);
    const LevelGeom& L = geom_for(m, width, height).g.lv[level];
    *lw = L.w; *lh = L.h;
    if ((int64_t)L.w * L.h > out_capacity) fail(SLIDEO_ERR_CAPACITY, "level needs %lld bytes", (long long)L.w * L.h);
    if (L.w > 0 && L.h > 0) {
        const uint8_t* src = (blurred ? S.d_blur.as<uint8_t>() : S.d_pyr.as<uint8_t>()) + L.ofs;
        HIP_CHECK(hipMemcpy2DAsync(out, L.w, src, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    API_CATCH(m)
}

int32_t slideo_small_image_bgr8(slideo_matcher* m, const uint8_t* bgr, int32_t width, int32_t height, int32_t stride_bytes,
                                uint8_t* out, int64_t out_capacity, int32_t* sw_out, int32_t* sh_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (!bgr || !out || !sw_out || !sh_out) fail(SLIDEO_ERR_INVALID_ARG, "null argument");
    validate_image(width, height, stride_bytes);
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    hipStream_t st = S.st;
    const size_t fb = (size_t)height * stride_bytes;
    stage_for_upload(m, fb);
    HIP_CHECK(hipMemcpyAsync(S.d_stage.p, bgr, fb, hipMemcpyHostToDevice, st));
    int sw = 0, sh = 0;
    run_small(m, S.d_stage.as<uint8_t>(), 1, width, height, stride_bytes, (int64_t)fb, sw, sh, st);
    *sw_out = sw; *sh_out = sh;
    if ((int64_t)sw * sh * 3 > out_capacity) fail(SLIDEO_ERR_CAPACITY, "small image needs %lld bytes", (long long)sw * sh * 3);
    HIP_CHECK(hipMemcpyAsync(out, m->d_small.p, (size_t)sw * sh * 3, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    API_CATCH(m)
}

}  // extern "C"
