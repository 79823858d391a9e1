// cv_math.hip.h — device restatements of OpenCV scalar math shared by the ORB and SIFT kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cfloat>

namespace slideo {

// [OCV A.5] fastAtan2 (f32): the odd degree-7 polynomial of core/src/mathfuncs_core.simd.hpp, degrees in [0, 360).
// fma = slideo_ocv_variants.atan 1: the Horner steps (and the final 90 - P c) contracted, as a compiler does when the
// scalar code is built for a baseline with FMA3
__device__ __forceinline__ float fast_atan2f_cv(float y, float x, bool fma) {
    const float s = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s,
                p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = fma ? __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(p7, c2, p5), c2, p3), c2, p1) * c
                : (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = fma ? __builtin_fmaf(-__builtin_fmaf(__builtin_fmaf(__builtin_fmaf(p7, c2, p5), c2, p3), c2, p1), c, 90.f)
                : 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

}  // namespace slideo
