/*
 * slideo_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the hot path of hediet/slideo's crates/matching-opencv
 * (per-frame ORB detect+describe -> exact Hamming kNN -> 5 % tolerance vote ->
 * RANSAC similarity -> re-projection similarity verdict -> timeline dedup).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (slideo_amd/) never does.
 *
 * PARITY UNPINNED.  The reference's arithmetic lives entirely in OpenCV 4.5.2
 * C++ (crate opencv 0.52.0, Cargo.lock:1723-1724; .github/workflows/ci.yml:18),
 * which is neither vendored under /root/reference nor installed here, and the
 * reference holds no test, golden vector or expected output for this path
 * (SURVEY.md F8/F9).  Control flow, thresholds and ordering below follow the
 * reference's Rust files line by line (cited as mo/<file>:<line>, mo/ =
 * crates/matching-opencv/src/); the primitive semantics follow OpenCV 4.5.2's
 * published algorithms as restated in SURVEY.md Appendix A (cited as [OCV A.n],
 * upstream file named).  The self-consistency constants of SURVEY.md Appendix B
 * (pattern SHA-256, quotas, umax, pyramid sizes, RNG draws) are asserted by
 * tests/test_oracle_constants.py; tests/test_oracle_crosscheck.py holds the
 * primitives to an independent implementation (scikit-image / SciPy known
 * answers in tests/golden/crosscheck.npz) — definitions, not OpenCV's rounding.
 *
 * Deliberate, documented departures from what the reference *runs*:
 *   - kNN is exact brute force, not FLANN-LSH (north_star);
 *   - every order the reference leaves to HashMap iteration / nth_element
 *      is canonicalised: keypoints (octave, y, x); pages tie-break
 *     by ascending page index;
 *   - the 4x4 damped normal equations of the LM refine are solved by Gaussian
 *     elimination with partial pivoting (OpenCV: DECOMP_EIG); equal to f64
 *     round-off.
 *
 * Every primitive whose exact OpenCV rounding could only be RECALLED (Appendix A,
 * confidence M / L) exists here in more than one form, selected by
 * slideo_config.ocv (slideo_ocv_variants, include/slideo_amd.h) — the same switches
 * the product's host tables read (slideo_amd/csrc/geom.h).  Value 0 of each is the
 * default; DESIGN.md section 5 says why.  tools/pin_opencv.py (run where cv2 4.5.2
 * exists) + tests/test_opencv_pin.py turn "unpinned" into a per-switch verdict in
 * one command; until that has happened this header keeps saying PARITY UNPINNED.
 *
 * The k-NN of match_frame is the cache-blocked form (knn_hamming_blocked: train rows
 * re-laid in 8-row qword-interleaved blocks, 16 queries x 512-row tiles, AVX-512
 * VPOPCNTDQ by runtime dispatch, scalar popcnt otherwise) — same results as the
 * plain so_knn_hamming (tests/test_oracle_primitives.py), so that bench.py's CPU leg
 * is not a strawman.
 *
 * North-star options WITHOUT a reference counterpart (off by default, same PARITY UNPINNED status — OpenCV 4.5.2
 * recalled, pinned by hand-computable cases and numpy / SciPy cross-checks in tests/test_oracle_{homography,sift,lsh}.py):
 *   - verify_model 1: find_homography (calib3d fundam.cpp / ptsetreg.cpp / levmarq.cpp, core lapack.cpp JacobiImpl_: RANSAC
 *     with getSubset / checkSubset, normalised DLT, refit + LMSolver) and warp_perspective_nn_bgr8 (imgwarp.cpp
 *     WarpPerspectiveInvoker); ocv.hdlt 1 = the 8x8 elimination form of the 4-point model, 2 = its closed form;
 *   - matcher 1: LshIdx, the candidate rule of FLANN's LshIndex as the reference configures it (mo/flann.rs:14-26);
 *   - sift_oracle.h: cv::SIFT::detectAndCompute (its own header lists what is restated and the two departures);
 *   - ratio_test, so_knn_l2_u8 (BFMatcher NORM_L2 on u8 descriptors).
 *
 * Build: see oracle/Makefile.  All floating point is compiled with
 * -ffp-contract=off so that results do not depend on FMA availability (the FMA
 * variants call fma() explicitly).
 */
#include "../include/slideo_amd.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------

// cvRound: round half to even (lrint under the default rounding mode).
static inline int cv_round(double v) { return (int)std::lrint(v); }
static inline int cv_floor(double v) { return (int)std::floor(v); }
static inline int cv_ceil(double v) { return (int)std::ceil(v); }

// [OCV A.7] cv::RNG — 64-bit multiply-with-carry (core/include/opencv2/core.hpp).
struct CvRng {
    uint64_t state;
    uint64_t mul;      // CV_RNG_COEFF = 4164903690 (slideo_ocv_variants.rng_mul)
    explicit CvRng(uint64_t s, uint32_t mul_ = 4164903690u) : state(s ? s : 0xffffffffULL), mul(mul_) {}
    uint32_t next() {
        state = (uint64_t)(uint32_t)state * mul + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a)) + a; }
};

struct Img8 {  // single-channel 8-bit image, tightly packed
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    Img8() {}
    Img8(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
    uint8_t* row(int y) { return d.data() + (size_t)y * w; }
    const uint8_t* row(int y) const { return d.data() + (size_t)y * w; }
};

static inline int reflect101(int p, int n) {
    // BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

// ---------------------------------------------------------------------------
// [OCV A.1] BGR -> gray, imgproc/src/color_rgb.simd.hpp RGB2Gray<uchar>
//   gray = (B*3735 + G*19235 + R*9798 + 2^14) >> 15
// ---------------------------------------------------------------------------
// ocv.gray: 0 = the Q15 coefficients above (4.x), 1 = Q14 1868/9617/4899 (2.4 / 3.x)
static void gray_bgr8(const uint8_t* bgr, int w, int h, int stride, Img8& out, int variant = 0) {
    out = Img8(w, h);
    const int cb = variant == 1 ? 1868 : 3735, cg = variant == 1 ? 9617 : 19235, cr = variant == 1 ? 4899 : 9798;
    const int sh = variant == 1 ? 14 : 15;
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = bgr + (size_t)y * stride;
        uint8_t* d = out.row(y);
        for (int x = 0; x < w; ++x) {
            int b = s[3 * x], g = s[3 * x + 1], r = s[3 * x + 2];
            d[x] = (uint8_t)((b * cb + g * cg + r * cr + (1 << (sh - 1))) >> sh);
        }
    }
}

// ---------------------------------------------------------------------------
// [OCV A.2] resize INTER_LINEAR_EXACT, imgproc/src/resize.cpp resize_bitExact:
// per-axis 8.8 fixed-point coefficients, exact integer accumulation,
// out = (sum_y cy * (sum_x cx * p) + 2^15) >> 16.
// ---------------------------------------------------------------------------
struct LinCoef { int ofs; int c0, c1; };  // value = c0*src[ofs] + c1*src[ofs+1]; weights sum to 256

// ocv.resize: 0 = cvRound (ties to even), 1 = floor(x + 0.5) (ties up)
static void linear_exact_coeffs(int ssize, int dsize, std::vector<LinCoef>& out, int variant = 0) {
    out.resize(dsize);
    double inv_scale = (double)dsize / (double)ssize;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dsize; ++d) {
        double f = scale * ((double)d + 0.5) - 0.5;
        int i = cv_floor(f);
        LinCoef c;
        if (i >= 0 && ssize > 1) {
            if (i < ssize - 1) {
                c.ofs = i;
                c.c1 = variant == 1 ? cv_floor((f - (double)i) * 256.0 + 0.5) : cv_round((f - (double)i) * 256.0);
                c.c0 = 256 - c.c1;
            } else {
                c.ofs = ssize - 1; c.c0 = 256; c.c1 = 0;   // right/bottom replicate
            }
        } else {
            c.ofs = 0; c.c0 = 256; c.c1 = 0;              // left/top replicate
        }
        out[d] = c;
    }
}

static void resize_linear_exact(const Img8& src, int dw, int dh, Img8& dst, int variant = 0) {
    dst = Img8(dw, dh);
    std::vector<LinCoef> cx, cy;
    linear_exact_coeffs(src.w, dw, cx, variant);
    linear_exact_coeffs(src.h, dh, cy, variant);
    std::vector<uint32_t> r0(dw), r1(dw);
    for (int y = 0; y < dh; ++y) {
        const LinCoef& yc = cy[y];
        const uint8_t* s0 = src.row(yc.ofs);
        const uint8_t* s1 = src.row(std::min(yc.ofs + 1, src.h - 1));
        uint8_t* d = dst.row(y);
        for (int x = 0; x < dw; ++x) {
            const LinCoef& xc = cx[x];
            int o1 = std::min(xc.ofs + 1, src.w - 1);
            uint32_t h0 = (uint32_t)xc.c0 * s0[xc.ofs] + (uint32_t)xc.c1 * s0[o1];
            uint32_t h1 = (uint32_t)xc.c0 * s1[xc.ofs] + (uint32_t)xc.c1 * s1[o1];
            uint32_t v = (uint32_t)yc.c0 * h0 + (uint32_t)yc.c1 * h1;
            d[x] = (uint8_t)((v + (1u << 15)) >> 16);
        }
    }
}

// ---------------------------------------------------------------------------
// [OCV A.2] pyramid geometry, features2d/src/orb.cpp detectAndCompute/getScale
// ---------------------------------------------------------------------------
static float level_scale(const slideo_config& c, int level) {
    return (float)std::pow((double)c.scale_factor, (double)level);
}

static void pyramid_sizes(int w, int h, const slideo_config& c, std::vector<int>& ws,
                          std::vector<int>& hs, std::vector<float>& scales) {
    ws.resize(c.nlevels); hs.resize(c.nlevels); scales.resize(c.nlevels);
    for (int l = 0; l < c.nlevels; ++l) {
        float s = level_scale(c, l);
        scales[l] = s;
        ws[l] = cv_round((float)w / s);
        hs[l] = cv_round((float)h / s);
    }
}

// [OCV A.4] per-level quotas, orb.cpp computeKeyPoints
static void level_quotas(const slideo_config& c, std::vector<int>& q) {
    q.assign(c.nlevels, 0);
    float factor = (float)(1.0 / (double)c.scale_factor);
    float ndesired = (float)c.nfeatures * (1.0f - factor) /
                     (1.0f - (float)std::pow((double)factor, (double)c.nlevels));
    int sum = 0;
    for (int l = 0; l < c.nlevels - 1; ++l) {
        q[l] = cv_round(ndesired);
        sum += q[l];
        ndesired *= factor;
    }
    q[c.nlevels - 1] = std::max(c.nfeatures - sum, 0);
}

// [OCV A.5] umax table, orb.cpp detectAndCompute
static void umax_table(int half_patch, std::vector<int>& umax) {
    umax.assign(half_patch + 2, 0);
    int v, v0;
    int vmax = cv_floor(half_patch * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(half_patch * std::sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v)
        umax[v] = cv_round(std::sqrt((double)half_patch * half_patch - (double)v * v));
    for (v = half_patch, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// [OCV A.7] BRIEF pattern for patchSize != 31, orb.cpp makeRandomPattern
static void brief_pattern(int patch_size, int npoints, std::vector<int32_t>& xy, uint32_t rng_mul = 4164903690u) {
    xy.resize((size_t)npoints * 2);
    CvRng rng(0x34985739, rng_mul);
    for (int i = 0; i < npoints; ++i) {
        xy[2 * i] = rng.uniform(-patch_size / 2, patch_size / 2 + 1);
        xy[2 * i + 1] = rng.uniform(-patch_size / 2, patch_size / 2 + 1);
    }
}

// ---------------------------------------------------------------------------
// [OCV A.3] FAST-9/16 with score, features2d/src/fast.cpp + fast_score.cpp
// ---------------------------------------------------------------------------
static const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},  {3, 0},  {3, -1},
                                   {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                   {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};

// Score of pixel (x,y): 0 if not a corner, else max over 9-arcs of the min
// |difference| minus 1 (== cornerScore<16>).  Requires 3 <= x < w-3, same y.
static inline int fast_score_at(const Img8& im, int x, int y, int t) {
    int v = im.row(y)[x];
    int d[16];
    for (int k = 0; k < 16; ++k) d[k] = v - im.row(y + kCircle[k][1])[x + kCircle[k][0]];
    int best = 0;  // max over arcs of min(d) (bright centre) and min(-d) (dark centre)
    for (int s = 0; s < 16; ++s) {
        int mn = d[s], mx = d[s];
        for (int j = 1; j < 9; ++j) {
            int dv = d[(s + j) & 15];
            mn = std::min(mn, dv);
            mx = std::max(mx, dv);
        }
        best = std::max(best, std::max(mn, -mx));
    }
    return best > t ? best - 1 : 0;
}

struct RawCorner { int x, y, score; };

// Score map over the scanned region [3,w-3)x[3,h-3), zeros elsewhere.
static void fast_score_map(const Img8& im, int t, Img8& sc) {
    sc = Img8(im.w, im.h);
    if (im.w < 7 || im.h < 7) return;
    for (int y = 3; y < im.h - 3; ++y) {
        uint8_t* s = sc.row(y);
        const uint8_t* c = im.row(y);
        const int st = im.w;
        for (int x = 3; x < im.w - 3; ++x) {
            // quick reject (fast.cpp tests the same antipodal pairs first): every arc of 9 contains one pixel of each antipodal
            // pair, so a pair whose two pixels are both within t of the centre rules the corner out.  Score 0 either way.
            const int v = c[x];
            auto far = [&](int dx, int dy) { const int p = c[x + dy * st + dx]; return p > v + t || p < v - t; };
            if (!(far(0, 3) || far(0, -3)) || !(far(3, 0) || far(-3, 0)) || !(far(2, 2) || far(-2, -2)) || !(far(2, -2) || far(-2, 2))) continue;
            s[x] = (uint8_t)fast_score_at(im, x, y, t);
        }
    }
}

// 3x3 non-max suppression: keep iff score strictly greater than all 8 neighbours.
static void fast_nms(const Img8& sc, std::vector<RawCorner>& out) {
    out.clear();
    for (int y = 3; y < sc.h - 3; ++y) {
        const uint8_t* p = sc.row(y - 1);
        const uint8_t* c = sc.row(y);
        const uint8_t* n = sc.row(y + 1);
        for (int x = 3; x < sc.w - 3; ++x) {
            int s = c[x];
            if (!s) continue;
            if (s > p[x - 1] && s > p[x] && s > p[x + 1] && s > c[x - 1] && s > c[x + 1] &&
                s > n[x - 1] && s > n[x] && s > n[x + 1])
                out.push_back({x, y, s});
        }
    }
}

// ---------------------------------------------------------------------------
// [OCV A.6] GaussianBlur 7x7 sigma 2 on 8-bit: fixed-point separable,
// imgproc/src/smooth.dispatch.cpp getGaussianKernelFixedPoint_ED + fixed-point
// hline/vline: out = (sum_j k_j * sum_i k_i * p + 2^15) >> 16, BORDER_REFLECT_101.
// ---------------------------------------------------------------------------
static void gauss_kernel_fixed(int n, double sigma, std::vector<int>& k) {
    std::vector<double> kd(n);
    double scale2x = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < n; ++i) {
        double x = i - (n - 1) * 0.5;
        kd[i] = std::exp(scale2x * x * x);
        sum += kd[i];
    }
    for (int i = 0; i < n; ++i) kd[i] *= 1.0 / sum;
    k.assign(n, 0);
    int n2 = n / 2;
    double err = 0;
    int64_t acc = 0;
    for (int i = 0; i < n2; ++i) {   // error diffusion, outer taps first
        double adj = kd[i] * 256.0 + err;
        int v0 = cv_round(adj);
        err = adj - (double)v0;
        k[i] = v0; k[n - 1 - i] = v0;
        acc += v0;
    }
    k[n2] = (int)(256 - 2 * acc);    // centre = remainder, kernel sums to exactly 256
}

// taps of ocv.blur 2: createSeparableLinearFilter before 4.2, kernel.convertTo(CV_32S, 256) = cvRound(k * 256)
static void gauss_kernel_q8_rounded(int n, double sigma, std::vector<int>& k) {
    std::vector<double> kd(n);
    double scale2x = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < n; ++i) { double x = i - (n - 1) * 0.5; kd[i] = std::exp(scale2x * x * x); sum += kd[i]; }
    k.assign(n, 0);
    for (int i = 0; i < n; ++i) k[i] = cv_round((double)(float)(kd[i] * (1.0 / sum)) * 256.0);
}

// f32 taps of ocv.blur 0 / 1: getGaussianKernel(n, sigma, CV_32F) = (float) of the f64 (softdouble) kernel, built as
// getGaussianKernelBitExact does: the n/2 outer values, sum = 2 * their sum + 1, each value times 1/sum
static void gauss_kernel_f32(int n, double sigma, std::vector<float>& k) {
    const int n2 = (n - 1) / 2;
    std::vector<double> v(n2 + 1);
    const double scale2x = -0.125 / (sigma * sigma);
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n2; ++i, x += 2) { v[i] = std::exp((double)(x * x) * scale2x); sum += v[i]; }
    sum *= 2.0; sum += 1.0;
    const double mul1 = 1.0 / sum;
    k.assign(n, 0.f);
    for (int i = 0; i < n2; ++i) { const double t = v[i] * mul1; k[i] = (float)t; k[n - 1 - i] = (float)t; }
    k[n2] = (float)(1.0 * mul1);
}

static inline float mad_f32(float a, float b, float c, bool fma) { return fma ? std::fmaf(a, b, c) : a * b + c; }

// ocv.blur (include/slideo_amd.h): 0 = sepFilter2D f32 with contracted products, 1 = f32 without, 2 = sepFilter2D Q8
// (rounded taps, saturating), 3 = GaussianBlur's bit-exact fixed-point path (error-diffused taps)
static void gaussian_blur7(const Img8& src, Img8& dst, int variant = 0) {
    int w = src.w, h = src.h;
    dst = Img8(w, h);
    if (variant == 0 || variant == 1) {
        // filter.simd.hpp RowFilter<uchar, float, RowNoVec>: s = k0 * p0; s += k_i * p_i (i = 1..6);
        // SymmColumnFilter<Cast<float, uchar>, ColumnNoVec>: s = k3 * c (+ delta 0); s += k_(3+j) * (r_(+j) + r_(-j)); cvRound, saturate
        const bool fma = variant == 0;
        std::vector<float> k;
        gauss_kernel_f32(7, 2.0, k);
        std::vector<float> tmp((size_t)w * h);
        for (int y = 0; y < h; ++y) {
            const uint8_t* s = src.row(y);
            float* t = tmp.data() + (size_t)y * w;
            for (int x = 0; x < w; ++x) {
                float a = k[0] * (float)s[reflect101(x - 3, w)];
                for (int i = 1; i < 7; ++i) a = mad_f32(k[i], (float)s[reflect101(x + i - 3, w)], a, fma);
                t[x] = a;
            }
        }
        for (int y = 0; y < h; ++y) {
            uint8_t* d = dst.row(y);
            const float* r[7];
            for (int i = 0; i < 7; ++i) r[i] = tmp.data() + (size_t)reflect101(y + i - 3, h) * w;
            for (int x = 0; x < w; ++x) {
                float a = mad_f32(k[3], r[3][x], 0.f, fma);
                for (int j = 1; j <= 3; ++j) a = mad_f32(k[3 + j], r[3 + j][x] + r[3 - j][x], a, fma);
                int iv = (int)std::lrintf(a);
                d[x] = (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
            }
        }
        return;
    }
    std::vector<int> k;
    if (variant == 2) gauss_kernel_q8_rounded(7, 2.0, k); else gauss_kernel_fixed(7, 2.0, k);
    std::vector<uint32_t> tmp((size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = src.row(y);
        uint32_t* t = tmp.data() + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            uint32_t a = 0;
            for (int i = 0; i < 7; ++i) a += (uint32_t)k[i] * s[reflect101(x + i - 3, w)];
            t[x] = a;
        }
    }
    for (int y = 0; y < h; ++y) {
        uint8_t* d = dst.row(y);
        const uint32_t* r[7];
        for (int i = 0; i < 7; ++i) r[i] = tmp.data() + (size_t)reflect101(y + i - 3, h) * w;
        for (int x = 0; x < w; ++x) {
            uint32_t a = 0;
            for (int i = 0; i < 7; ++i) a += (uint32_t)k[i] * r[i][x];
            a = (a + (1u << 15)) >> 16;
            d[x] = (uint8_t)(a > 255u ? 255u : a);          // (only the 257-sum taps of variant 2 can exceed 255)
        }
    }
}

// ---------------------------------------------------------------------------
// [OCV A.5] fastAtan2, core/src/mathfuncs_core.simd.hpp atan_f32 (all f32)
// ---------------------------------------------------------------------------
// ocv.atan: 0 = plain multiplies and adds, 1 = the Horner steps contracted to fma
static float fast_atan2(float y, float x, int variant = 0) {
    const float s = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s,
                p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const bool fma = variant == 1;
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = mad_f32(mad_f32(mad_f32(p7, c2, p5, fma), c2, p3, fma), c2, p1, fma) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        // (with contraction GCC folds the final multiply into the subtraction: 90 - P*c = fma(-P, c, 90))
        const float P = mad_f32(mad_f32(mad_f32(p7, c2, p5, fma), c2, p3, fma), c2, p1, fma);
        a = fma ? std::fmaf(-P, c, 90.f) : 90.f - P * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// [OCV A.5] intensity-centroid angle, orb.cpp ICAngles (unblurred level image)
static float ic_angle(const Img8& im, int cx, int cy, const std::vector<int>& umax, int half, int atan_variant = 0) {
    int m01 = 0, m10 = 0;
    const uint8_t* c = im.row(cy) + cx;
    int step = im.w;
    for (int u = -half; u <= half; ++u) m10 += u * c[u];
    for (int v = 1; v <= half; ++v) {
        int vsum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int vp = c[u + v * step], vm = c[u - v * step];
            vsum += (vp - vm);
            m10 += u * (vp + vm);
        }
        m01 += v * vsum;
    }
    return fast_atan2((float)m01, (float)m10, atan_variant);
}

// ---------------------------------------------------------------------------
// ORB detect + describe (mo/feature_extractor.rs:29-46 -> [OCV A.1-A.7])
// Output canonical order (octave, y, x) — SURVEY F11.
// ---------------------------------------------------------------------------
struct OrbResult {
    std::vector<slideo_keypoint> kp;
    std::vector<uint8_t> desc;  // 32 bytes each
};

static bool config_supported(const slideo_config& c) {
    int half = c.patch_size / 2;
    int desc_r = cv_ceil(half * std::sqrt(2.0));
    // keypoints are >= edge_threshold from the level edge; the IC disc (radius half) and the
    // rotated BRIEF samples (radius desc_r) plus the 7x7 blur support must stay inside the level.
    return c.nlevels >= 1 && c.nlevels <= 16 && c.nfeatures > 0 && c.patch_size != 31 &&
           c.patch_size >= 2 && c.edge_threshold >= desc_r + 3 && c.edge_threshold >= half &&
           c.edge_threshold >= 4 && c.scale_factor > 1.0f && c.fast_threshold >= 1 &&
           c.fast_threshold < 255 && c.knn_k >= 1 && c.knn_k <= 64 &&
           c.ratio_test >= 0.f && !(c.ratio_test > 0.f && c.knn_k < 2) &&
           c.ocv.gray >= 0 && c.ocv.gray <= 1 && c.ocv.blur >= 0 && c.ocv.blur <= 3 && c.ocv.resize >= 0 && c.ocv.resize <= 1 &&
           c.ocv.atan >= 0 && c.ocv.atan <= 1 && c.ocv.warp >= 0 && c.ocv.warp <= 1 && c.ocv.area >= 0 && c.ocv.area <= 1 &&
           c.ocv.lm >= 0 && c.ocv.lm <= 1 && c.ocv.hdlt >= 0 && c.ocv.hdlt <= 2 &&
           c.verify_model >= 0 && c.verify_model <= 1 && c.matcher >= 0 && c.matcher <= 1 && c.verdict_rule >= 0 && c.verdict_rule <= 1 &&
           (c.matcher == 0 || (c.lsh_tables >= 1 && c.lsh_tables <= 8 && c.lsh_key_bits >= 1 && c.lsh_key_bits <= 16 && c.lsh_multi_probe >= 0 &&
                               c.lsh_multi_probe <= 2 && !(c.ratio_test > 0.f)));
}

struct Pyramid {
    std::vector<Img8> level;    // unblurred
    std::vector<Img8> blurred;  // after GaussianBlur
    std::vector<float> scale;
};

static void build_pyramid(const uint8_t* bgr, int w, int h, int stride, const slideo_config& c,
                          Pyramid& p, bool with_blur) {
    std::vector<int> ws, hs;
    pyramid_sizes(w, h, c, ws, hs, p.scale);
    p.level.resize(c.nlevels);
    gray_bgr8(bgr, w, h, stride, p.level[0], c.ocv.gray);
    for (int l = 1; l < c.nlevels; ++l) {
        if (ws[l] < 1 || hs[l] < 1) { p.level[l] = Img8(0, 0); continue; }
        resize_linear_exact(p.level[l - 1], ws[l], hs[l], p.level[l], c.ocv.resize);  // progressive
    }
    if (with_blur) {
        p.blurred.resize(c.nlevels);
        for (int l = 0; l < c.nlevels; ++l)
            if (p.level[l].w > 0) gaussian_blur7(p.level[l], p.blurred[l], c.ocv.blur);
    }
}

static void orb_detect_describe(const uint8_t* bgr, int w, int h, int stride,
                                const slideo_config& c, OrbResult& out) {
    out.kp.clear(); out.desc.clear();
    Pyramid pyr;
    build_pyramid(bgr, w, h, stride, c, pyr, true);
    std::vector<int> quota, umax;
    level_quotas(c, quota);
    const int half = c.patch_size / 2;
    umax_table(half, umax);
    std::vector<int32_t> pat;
    brief_pattern(c.patch_size, 512, pat, c.ocv.rng_mul);
    const int edge = c.edge_threshold;

    for (int l = 0; l < c.nlevels; ++l) {
        const Img8& im = pyr.level[l];
        if (im.w <= 0) continue;
        // FAST + NMS on the level image (no border use)
        Img8 sc;
        fast_score_map(im, c.fast_threshold, sc);
        std::vector<RawCorner> corners;
        fast_nms(sc, corners);
        // runByImageBorder(edge): clears all if the level is too small
        std::vector<RawCorner> kept;
        if (!(im.w <= edge * 2 || im.h <= edge * 2)) {
            for (const RawCorner& r : corners)
                if (r.x >= edge && r.x < im.w - edge && r.y >= edge && r.y < im.h - edge)
                    kept.push_back(r);
        }
        // retainBest(n): keep the n best by response plus every tie with the n-th
        int n = quota[l];
        if ((int)kept.size() > n) {
            if (n == 0) kept.clear();
            else {
                std::vector<int> sc_sorted;
                sc_sorted.reserve(kept.size());
                for (const RawCorner& r : kept) sc_sorted.push_back(r.score);
                std::nth_element(sc_sorted.begin(), sc_sorted.begin() + (n - 1), sc_sorted.end(),
                                 std::greater<int>());
                int thr = sc_sorted[n - 1];
                std::vector<RawCorner> k2;
                for (const RawCorner& r : kept) if (r.score >= thr) k2.push_back(r);
                kept.swap(k2);
            }
        }
        // `kept` is already in (y, x) order (row-major scan) == canonical order.
        const float sf = pyr.scale[l];
        const Img8& bl = pyr.blurred[l];
        for (const RawCorner& r : kept) {
            slideo_keypoint k;
            k.octave = l;
            k.size = (float)c.patch_size * sf;
            k.response = (float)r.score;
            k.angle = ic_angle(im, r.x, r.y, umax, half, c.ocv.atan);
            k.x = (float)r.x * sf;
            k.y = (float)r.y * sf;
            // descriptor on the blurred level, orb.cpp computeOrbDescriptors
            float inv = 1.f / sf;
            float ang = k.angle * (float)(3.14159265358979323846 / 180.f);
            float a = (float)std::cos((double)ang), b = (float)std::sin((double)ang);
            int cx = cv_round(k.x * inv), cy = cv_round(k.y * inv);
            uint8_t d[32];
            for (int j = 0; j < 32; ++j) {
                int byte = 0;
                for (int bit = 0; bit < 8; ++bit) {
                    int i0 = 16 * j + 2 * bit, i1 = i0 + 1;
                    float x0 = (float)pat[2 * i0] * a - (float)pat[2 * i0 + 1] * b;
                    float y0 = (float)pat[2 * i0] * b + (float)pat[2 * i0 + 1] * a;
                    float x1 = (float)pat[2 * i1] * a - (float)pat[2 * i1 + 1] * b;
                    float y1 = (float)pat[2 * i1] * b + (float)pat[2 * i1 + 1] * a;
                    int t0 = bl.row(cy + cv_round(y0))[cx + cv_round(x0)];
                    int t1 = bl.row(cy + cv_round(y1))[cx + cv_round(x1)];
                    byte |= (t0 < t1) << bit;
                }
                d[j] = (uint8_t)byte;
            }
            out.kp.push_back(k);
            out.desc.insert(out.desc.end(), d, d + 32);
        }
    }
}

// ---------------------------------------------------------------------------
// [OCV A.8] exact Hamming kNN == BFMatcher(NORM_HAMMING).knnMatch
// (features2d/src/matchers.cpp, core/src/batch_distance.cpp): ascending by
// distance, ties to the lower train row.
// ---------------------------------------------------------------------------
static inline int hamming256(const uint8_t* a, const uint8_t* b) {
    uint64_t x[4], y[4];
    std::memcpy(x, a, 32); std::memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

static void knn_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int k,
                        int32_t* idx, uint16_t* dist) {
    std::vector<int> bd(k), bi(k);
    for (int i = 0; i < nq; ++i) {
        int cnt = 0;
        const uint8_t* qi = q + (size_t)i * 32;
        for (int j = 0; j < nt; ++j) {
            int d = hamming256(qi, t + (size_t)j * 32);
            if (cnt == k && d >= bd[k - 1]) continue;  // strict: later equal distance never enters
            int p = cnt < k ? cnt++ : k - 1;
            while (p > 0 && bd[p - 1] > d) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
            bd[p] = d; bi[p] = j;
        }
        for (int r = 0; r < k; ++r) {
            idx[(size_t)i * k + r] = r < cnt ? bi[r] : -1;
            dist[(size_t)i * k + r] = r < cnt ? (uint16_t)bd[r] : (uint16_t)65535;
        }
    }
}

// ---------------------------------------------------------------------------
// The same exact k-NN, cache-blocked and vectorised — what the frame path and bench.py's cpu_baseline leg run, so that the
// CPU number beside the GPU's is not a strawman (the plain loop above streams the 16.5 MB train matrix once per
// QUERY and scaled 6.7x on 256 threads).  Train rows are re-laid in blocks of 8 rows, qword-interleaved
// (block b = [w = 0..3][row = 0..7] u64: one 64-byte vector per descriptor qword), so that for a broadcast query qword the
// eight distances of a block are 4 x (xor, popcount, add) on full vectors with no horizontal step; queries are taken 16 at a
// time against train tiles of 512 rows (16 KB: the tile stays in L1 while the 16 queries sweep it, and the whole matrix is
// read once per 16 queries).  AVX-512 VPOPCNTDQ when the CPU has it (runtime dispatch), scalar popcnt on the same layout
// otherwise.  Rows are visited in ascending order per query and the test is strict, so the result equals knn_hamming's bit
// for bit (tests/test_oracle_primitives.py).
// ---------------------------------------------------------------------------
static void knn_block_train(const uint8_t* t, int nt, std::vector<uint64_t>& tb) {
    const int nb = (nt + 7) / 8;
    tb.assign((size_t)nb * 32, ~0ull);                      // pad rows: all ones (never selected: excluded by row index)
    for (int r = 0; r < nt; ++r) {
        uint64_t w[4];
        std::memcpy(w, t + (size_t)r * 32, 32);
        for (int k = 0; k < 4; ++k) tb[(size_t)(r >> 3) * 32 + k * 8 + (r & 7)] = w[k];
    }
}

static inline void knn_offer(int d, int row, int k, int& cnt, int* bd, int* bi) {
    if (cnt == k && d >= bd[k - 1]) return;                 // strict: a later equal distance never enters
    int p = cnt < k ? cnt++ : k - 1;
    while (p > 0 && bd[p - 1] > d) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
    bd[p] = d; bi[p] = row;
}

static void knn_scan_scalar(const uint64_t* tb, int b0, int b1, int nt, const uint64_t* q, int k, int& cnt, int* bd, int* bi) {
    for (int b = b0; b < b1; ++b) {
        const uint64_t* T = tb + (size_t)b * 32;
        for (int r = 0; r < 8; ++r) {
            const int row = b * 8 + r;
            if (row >= nt) break;
            const int d = __builtin_popcountll(T[r] ^ q[0]) + __builtin_popcountll(T[8 + r] ^ q[1]) +
                          __builtin_popcountll(T[16 + r] ^ q[2]) + __builtin_popcountll(T[24 + r] ^ q[3]);
            knn_offer(d, row, k, cnt, bd, bi);
        }
    }
}

#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx512f,avx512vpopcntdq")))
static void knn_scan_avx512(const uint64_t* tb, int b0, int b1, int nt, const uint64_t* q, int k, int& cnt, int* bd, int* bi) {
    const __m512i q0 = _mm512_set1_epi64((long long)q[0]), q1 = _mm512_set1_epi64((long long)q[1]);
    const __m512i q2 = _mm512_set1_epi64((long long)q[2]), q3 = _mm512_set1_epi64((long long)q[3]);
    for (int b = b0; b < b1; ++b) {
        const __m512i* T = reinterpret_cast<const __m512i*>(tb + (size_t)b * 32);
        __m512i d = _mm512_popcnt_epi64(_mm512_xor_si512(_mm512_loadu_si512(T), q0));
        d = _mm512_add_epi64(d, _mm512_popcnt_epi64(_mm512_xor_si512(_mm512_loadu_si512(T + 1), q1)));
        d = _mm512_add_epi64(d, _mm512_popcnt_epi64(_mm512_xor_si512(_mm512_loadu_si512(T + 2), q2)));
        d = _mm512_add_epi64(d, _mm512_popcnt_epi64(_mm512_xor_si512(_mm512_loadu_si512(T + 3), q3)));
        const long long thr = cnt == k ? bd[k - 1] : 257;
        __mmask8 m = _mm512_cmplt_epu64_mask(d, _mm512_set1_epi64(thr));
        if (!m) continue;
        alignas(64) uint64_t dv[8];
        _mm512_store_si512(dv, d);
        for (int r = 0; r < 8; ++r)                       // ascending rows; the threshold tightens inside the block
            if (((m >> r) & 1) && b * 8 + r < nt) knn_offer((int)dv[r], b * 8 + r, k, cnt, bd, bi);
    }
}
static bool cpu_has_vpopcntdq() {
    static const bool have = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vpopcntdq");
    return have;
}
#else
static bool cpu_has_vpopcntdq() { return false; }
#endif

// tb: knn_block_train(train); same outputs as knn_hamming
static void knn_hamming_blocked(const uint8_t* q, int nq, const uint64_t* tb, int nt, int k, int32_t* idx, uint16_t* dist) {
    constexpr int QB = 16, TILE_BLOCKS = 64;               // 16 queries x 512 train rows
    const int nb = (nt + 7) / 8;
    const bool simd = cpu_has_vpopcntdq();
    std::vector<int> bd((size_t)QB * k), bi((size_t)QB * k);
    for (int q0 = 0; q0 < nq; q0 += QB) {
        const int qn = std::min(QB, nq - q0);
        int cnt[QB] = {0};
        uint64_t qw[QB][4];
        for (int i = 0; i < qn; ++i) std::memcpy(qw[i], q + (size_t)(q0 + i) * 32, 32);
        for (int b0 = 0; b0 < nb; b0 += TILE_BLOCKS) {
            const int b1 = std::min(nb, b0 + TILE_BLOCKS);
            for (int i = 0; i < qn; ++i) {
#if defined(__x86_64__)
                if (simd) { knn_scan_avx512(tb, b0, b1, nt, qw[i], k, cnt[i], &bd[(size_t)i * k], &bi[(size_t)i * k]); continue; }
#endif
                knn_scan_scalar(tb, b0, b1, nt, qw[i], k, cnt[i], &bd[(size_t)i * k], &bi[(size_t)i * k]);
            }
        }
        for (int i = 0; i < qn; ++i)
            for (int r = 0; r < k; ++r) {
                idx[(size_t)(q0 + i) * k + r] = r < cnt[i] ? bi[(size_t)i * k + r] : -1;
                dist[(size_t)(q0 + i) * k + r] = r < cnt[i] ? (uint16_t)bd[(size_t)i * k + r] : (uint16_t)65535;
            }
    }
}

// ---------------------------------------------------------------------------
// [OCV A.9] estimateAffinePartial2D (RANSAC + LM refine), calib3d/src/ptsetreg.cpp
// via mo/image_utils.rs:45-60.  from = slide points, to = frame points.
// Returns found; M = 2x3 row-major (f64); mask[n] inlier flags.
// ---------------------------------------------------------------------------
struct P2f { float x, y; };

static void similarity_from_2(const P2f* f, const P2f* t, double M[6]) {
    double x1 = f[0].x, y1 = f[0].y, x2 = f[1].x, y2 = f[1].y;
    double X1 = t[0].x, Y1 = t[0].y, X2 = t[1].x, Y2 = t[1].y;
    double d = 1. / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
    double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
    double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
    double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) -
                     (X1 * x2 - X2 * x1) * (x1 - x2));
    double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) -
                     (Y1 * y2 - Y2 * y1) * (y1 - y2));
    M[0] = M[4] = S0; M[1] = -S1; M[2] = S2; M[3] = S1; M[5] = S3;
}

static int find_inliers(const P2f* from, const P2f* to, int n, const double M[6], float thr2,
                        uint8_t* mask) {
    float F0 = (float)M[0], F1 = (float)M[1], F2 = (float)M[2];
    float F3 = (float)M[3], F4 = (float)M[4], F5 = (float)M[5];
    int good = 0;
    for (int i = 0; i < n; ++i) {
        float a = F0 * from[i].x + F1 * from[i].y + F2 - to[i].x;
        float b = F3 * from[i].x + F4 * from[i].y + F5 - to[i].y;
        float e = a * a + b * b;
        int f = e <= thr2;   // NaN -> 0
        mask[i] = (uint8_t)f;
        good += f;
    }
    return good;
}

static int ransac_update_iters(double p, double ep, int model_points, int max_iters) {
    p = std::max(p, 0.); p = std::min(p, 1.);
    ep = std::max(ep, 0.); ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round(num / denom);
}

// Solve the 4x4 system A x = b (Gaussian elimination, partial pivoting).
static bool solve4(const double Ain[16], const double bin[4], double x[4]) {
    double A[4][5];
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) A[i][j] = Ain[i * 4 + j]; A[i][4] = bin[i]; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(A[r][c]) > std::fabs(A[p][c])) p = r;
        if (A[p][c] == 0.0) return false;
        if (p != c) for (int j = 0; j < 5; ++j) std::swap(A[p][j], A[c][j]);
        for (int r = c + 1; r < 4; ++r) {
            double f = A[r][c] / A[c][c];
            for (int j = c; j < 5; ++j) A[r][j] -= f * A[c][j];
        }
    }
    for (int i = 3; i >= 0; --i) {
        double s = A[i][4];
        for (int j = i + 1; j < 4; ++j) s -= A[i][j] * x[j];
        x[i] = s / A[i][i];
    }
    return true;
}

// core/src/lapack.cpp: the hypot OpenCV's Jacobi sweep uses (its own template, not libm's)
static inline double hypot_cv(double a, double b) {
    a = std::fabs(a); b = std::fabs(b);
    if (a > b) { b /= a; return a * std::sqrt(1 + b * b); }
    if (b > 0) { a /= b; return b * std::sqrt(1 + a * a); }
    return 0;
}

// core/src/lapack.cpp JacobiImpl_<double> — what cv::eigen runs on a symmetric matrix and what cv::solve(DECOMP_EIG)
// builds on: pivot = the largest off-diagonal element of the upper triangle (tracked per row / per column in
// indR / indC), at most n*n*30 rotations, stop when |pivot| <= DBL_EPSILON, eigenvalues sorted descending with
// their eigenvectors (the ROWS of V).  A (n x n, row-major) is destroyed; only its upper triangle is read.
// Returns the number of rotations.
static int jacobi_eig_n(double* A, int n, double* W, double* V) {
    const double eps = DBL_EPSILON;
    int indR[16], indC[16];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
    double mv = 0;
    int i, k, m, iters = 0;
    for (k = 0; k < n; ++k) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) {
            for (m = k + 1, mv = std::fabs(A[n * k + m]), i = k + 2; i < n; ++i) { double v = std::fabs(A[n * k + i]); if (mv < v) mv = v, m = i; }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = std::fabs(A[k]), i = 1; i < k; ++i) { double v = std::fabs(A[n * i + k]); if (mv < v) mv = v, m = i; }
            indC[k] = m;
        }
    }
    if (n > 1) for (iters = 0; iters < n * n * 30; ++iters) {
        for (k = 0, mv = std::fabs(A[indR[0]]), i = 1; i < n - 1; ++i) { double v = std::fabs(A[n * i + indR[i]]); if (mv < v) mv = v, k = i; }
        int l = indR[k];
        for (i = 1; i < n; ++i) { double v = std::fabs(A[n * indC[i] + i]); if (mv < v) mv = v, k = indC[i], l = i; }
        double p = A[n * k + l];
        if (std::fabs(p) <= eps) break;
        double y = (W[l] - W[k]) * 0.5;
        double t = std::fabs(y) + hypot_cv(p, y);
        double sn = hypot_cv(p, t);
        double c = t / sn;
        sn = p / sn; t = (p / t) * p;
        if (y < 0) sn = -sn, t = -t;
        A[n * k + l] = 0;
        W[k] -= t; W[l] += t;
        double a0, b0;
#define SO_ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * sn, v1 = a0 * sn + b0 * c
        for (i = 0; i < k; ++i) SO_ROT(A[n * i + k], A[n * i + l]);
        for (i = k + 1; i < l; ++i) SO_ROT(A[n * k + i], A[n * i + l]);
        for (i = l + 1; i < n; ++i) SO_ROT(A[n * k + i], A[n * l + i]);
        for (i = 0; i < n; ++i) SO_ROT(V[n * k + i], V[n * l + i]);
#undef SO_ROT
        for (int j = 0; j < 2; ++j) {
            int idx = j == 0 ? k : l;
            if (idx < n - 1) {
                for (m = idx + 1, mv = std::fabs(A[n * idx + m]), i = idx + 2; i < n; ++i) { double v = std::fabs(A[n * idx + i]); if (mv < v) mv = v, m = i; }
                indR[idx] = m;
            }
            if (idx > 0) {
                for (m = 0, mv = std::fabs(A[idx]), i = 1; i < idx; ++i) { double v = std::fabs(A[n * i + idx]); if (mv < v) mv = v, m = i; }
                indC[idx] = m;
            }
        }
    }
    for (k = 0; k < n - 1; ++k) {
        m = k;
        for (i = k + 1; i < n; ++i) if (W[m] < W[i]) m = i;
        if (k != m) { std::swap(W[m], W[k]); for (i = 0; i < n; ++i) std::swap(V[n * m + i], V[n * k + i]); }
    }
    return iters;
}

// ocv.lm 1: cv::solve(A, b, DECOMP_EIG) — the Jacobi sweep above followed by SVBkSb:
// x = sum_i (e_i . b) / w_i * e_i over the w_i above 2 eps * sum(w).
template <int N>
static bool solve_eig_n(const double* Ain, const double* bin, double* x) {
    double A[N * N], W[N], V[N * N];
    std::memcpy(A, Ain, sizeof(A));
    jacobi_eig_n(A, N, W, V);
    double thr = 0;
    for (int i = 0; i < N; ++i) thr += W[i];
    thr *= DBL_EPSILON * 2;
    for (int j = 0; j < N; ++j) x[j] = 0;
    for (int i = 0; i < N; ++i) {
        if (std::fabs(W[i]) <= thr) continue;
        double sdot = 0;
        for (int j = 0; j < N; ++j) sdot += V[N * i + j] * bin[j];
        sdot *= 1 / W[i];
        for (int j = 0; j < N; ++j) x[j] = x[j] + sdot * V[N * i + j];
    }
    return true;
}

static bool solve4_eig(const double Ain[16], const double bin[4], double x[4]) { return solve_eig_n<4>(Ain, bin, x); }

static bool solve4_v(const double A[16], const double b[4], double x[4], int lm_variant) {
    return lm_variant == 1 ? solve4_eig(A, b, x) : solve4(A, b, x);
}

// residuals + (optionally) J^T J and J^T r for h = (a, b, tx, ty); returns |r|^2
static double lm_eval(const P2f* src, const P2f* dst, int n, const double h[4], double* A,
                      double* v, double* rinf) {
    double S = 0, ri = 0;
    if (A) { std::fill(A, A + 16, 0.0); std::fill(v, v + 4, 0.0); }
    for (int i = 0; i < n; ++i) {
        double Mx = src[i].x, My = src[i].y;
        double xi = h[0] * Mx - h[1] * My + h[2];
        double yi = h[1] * Mx + h[0] * My + h[3];
        double ex = xi - dst[i].x, ey = yi - dst[i].y;
        S += ex * ex; S += ey * ey;
        ri = std::max(ri, std::max(std::fabs(ex), std::fabs(ey)));
        if (A) {
            const double J0[4] = {Mx, -My, 1., 0.}, J1[4] = {My, Mx, 0., 1.};
            for (int a = 0; a < 4; ++a) {
                for (int b = 0; b < 4; ++b) A[a * 4 + b] += J0[a] * J0[b] + J1[a] * J1[b];
                v[a] += J0[a] * ex + J1[a] * ey;
            }
        }
    }
    if (rinf) *rinf = ri;
    return S;
}

// [OCV] calib3d/src/levmarq.cpp LMSolverImpl::run, 4 parameters, maxIters iterations
static void lm_refine(const P2f* src, const P2f* dst, int n, double h[4], int max_iters, int lm_variant = 0) {
    const double eps = (double)FLT_EPSILON;
    double x[4] = {h[0], h[1], h[2], h[3]}, xd[4], A[16], v[4], D[4], d[4], Ap[16], rinf = 0;
    double S = lm_eval(src, dst, n, x, A, v, &rinf);
    for (int i = 0; i < 4; ++i) D[i] = A[i * 4 + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        std::memcpy(Ap, A, sizeof(A));
        for (int i = 0; i < 4; ++i) Ap[i * 4 + i] += lambda * D[i];
        if (!solve4_v(Ap, v, d, lm_variant)) { std::fill(d, d + 4, 0.0); }
        for (int i = 0; i < 4; ++i) xd[i] = x[i] - d[i];
        double Sd = lm_eval(src, dst, n, xd, nullptr, nullptr, nullptr);
        // temp_d = -A d + 2 v ; dS = d . temp_d
        double dS = 0;
        for (int i = 0; i < 4; ++i) {
            double t = 2 * v[i];
            for (int j = 0; j < 4; ++j) t -= A[i * 4 + j] * d[j];
            dS += d[i] * t;
        }
        double R = (S - Sd) / (std::fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double t = 0;
            for (int i = 0; i < 4; ++i) t += d[i] * v[i];
            double nu = (Sd - S) / (std::fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = std::min(std::max(nu, 2.), 10.);
            if (lambda == 0) {
                // invert(A): max |diag(A^-1)| via four solves
                double maxval = DBL_EPSILON;
                for (int i = 0; i < 4; ++i) {
                    double e[4] = {0, 0, 0, 0}, col[4];
                    e[i] = 1;
                    if (solve4_v(A, e, col, lm_variant)) maxval = std::max(maxval, std::fabs(col[i]));
                }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            std::memcpy(x, xd, sizeof(x));
            lm_eval(src, dst, n, x, A, v, &rinf);
        }
        iter++;
        double dinf = 0;
        for (int i = 0; i < 4; ++i) dinf = std::max(dinf, std::fabs(d[i]));
        bool proceed = iter < max_iters && dinf >= eps && rinf >= eps;
        if (!proceed) break;
    }
    std::memcpy(h, x, sizeof(x));
}

static bool estimate_affine_partial(const P2f* from, const P2f* to, int count,
                                    const slideo_config& c, double M[6], uint8_t* mask,
                                    int* iters_run) {
    const int model_points = 2;
    std::fill(M, M + 6, 0.0);
    std::fill(mask, mask + count, (uint8_t)0);
    if (iters_run) *iters_run = 0;
    if (count < model_points) return false;
    bool result = false;
    if (count == model_points) {
        similarity_from_2(from, to, M);
        mask[0] = mask[1] = 1;
        result = true;
    } else {
        const float thr2 = (float)(c.ransac_threshold * c.ransac_threshold);
        int niters = std::max(c.ransac_max_iters, 1), max_good = 0, iter;
        CvRng rng((uint64_t)-1, c.ocv.rng_mul);   // fresh per call
        std::vector<uint8_t> cur(count);
        double Mi[6];
        for (iter = 0; iter < niters; ++iter) {
            int i0 = rng.uniform(0, count), i1;
            for (i1 = rng.uniform(0, count); i1 == i0; i1 = rng.uniform(0, count)) {}
            P2f f[2] = {from[i0], from[i1]}, t[2] = {to[i0], to[i1]};
            similarity_from_2(f, t, Mi);
            int good = find_inliers(from, to, count, Mi, thr2, cur.data());
            if (good > std::max(max_good, model_points - 1)) {
                std::memcpy(mask, cur.data(), count);
                std::memcpy(M, Mi, sizeof(Mi));
                max_good = good;
                niters = ransac_update_iters(c.ransac_confidence, (double)(count - good) / count,
                                             model_points, niters);
            }
        }
        if (iters_run) *iters_run = iter;
        result = max_good > 0;
        if (!result) { std::fill(M, M + 6, 0.0); std::fill(mask, mask + count, (uint8_t)0); }
    }
    if (result && count > 2 && c.refine_iters > 0) {
        std::vector<P2f> s, d;
        for (int i = 0; i < count; ++i) if (mask[i]) { s.push_back(from[i]); d.push_back(to[i]); }
        if (!s.empty()) {
            double h[4] = {M[0], M[3], M[2], M[5]};
            lm_refine(s.data(), d.data(), (int)s.size(), h, c.refine_iters, c.ocv.lm);
            M[0] = M[4] = h[0]; M[1] = -h[1]; M[2] = h[2]; M[3] = h[1]; M[5] = h[3];
        }
    }
    return result;
}

// ---------------------------------------------------------------------------
// verify_model 1 — cv::findHomography(from, to, RANSAC, thr, mask, maxIters, confidence), calib3d/src/fundam.cpp +
// ptsetreg.cpp + levmarq.cpp of OpenCV 4.5.2, RECALLED (the reference never fits a homography — SURVEY F4 — so this
// row of SURVEY 8(f) N4 has no reference counterpart and no call site to anchor on; the parity target of the HIP path
// is this restatement).  H maps from (slide) -> to (frame), 3x3 row-major, H[8] = 1.
// ---------------------------------------------------------------------------

// HomographyEstimatorCallback::runKernel.  Returns the number of models (0 when a coordinate has no spread).
// rotations (may be null) receives the Jacobi rotation count.
static int homography_dlt(const P2f* M, const P2f* m, int count, double H[9], int* rotations = nullptr) {
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; ++i) { cmx += m[i].x; cmy += m[i].y; cMx += M[i].x; cMy += M[i].y; }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; ++i) {
        smx += std::fabs(m[i].x - cmx); smy += std::fabs(m[i].y - cmy);
        sMx += std::fabs(M[i].x - cMx); sMy += std::fabs(M[i].y - cMy);
    }
    if (std::fabs(smx) < DBL_EPSILON || std::fabs(smy) < DBL_EPSILON || std::fabs(sMx) < DBL_EPSILON || std::fabs(sMy) < DBL_EPSILON) return 0;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    double LtL[81];
    std::fill(LtL, LtL + 81, 0.0);
    for (int i = 0; i < count; ++i) {
        double x = (m[i].x - cmx) * smx, y = (m[i].y - cmy) * smy;
        double X = (M[i].x - cMx) * sMx, Y = (M[i].y - cMy) * sMy;
        double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; ++j)
            for (int k = j; k < 9; ++k) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; ++j) for (int k = 0; k < j; ++k) LtL[j * 9 + k] = LtL[k * 9 + j];      // completeSymm
    double W[9], V[81];
    int rot = jacobi_eig_n(LtL, 9, W, V);
    if (rotations) *rotations = rot;
    const double* h = V + 72;                           // eigenvector of the smallest eigenvalue
    // Htemp = invHnorm * H0, invHnorm = [1/sm.x 0 cm.x; 0 1/sm.y cm.y; 0 0 1]
    const double ix = 1. / smx, iy = 1. / smy;
    double T[9];
    for (int j = 0; j < 3; ++j) { T[j] = ix * h[j] + cmx * h[6 + j]; T[3 + j] = iy * h[3 + j] + cmy * h[6 + j]; T[6 + j] = h[6 + j]; }
    // H0 = Htemp * Hnorm2, Hnorm2 = [sM.x 0 -cM.x sM.x; 0 sM.y -cM.y sM.y; 0 0 1]
    const double n2 = -cMx * sMx, n5 = -cMy * sMy;
    double R[9];
    for (int r = 0; r < 3; ++r) { R[3 * r] = T[3 * r] * sMx; R[3 * r + 1] = T[3 * r + 1] * sMy; R[3 * r + 2] = T[3 * r] * n2 + T[3 * r + 1] * n5 + T[3 * r + 2]; }
    const double sc = 1. / R[8];                        // convertTo(model, type, 1 / H0(2,2))
    for (int j = 0; j < 9; ++j) H[j] = R[j] * sc;
    return 1;
}

// Gaussian elimination with partial pivoting, N x N (the LM solver's default form here, ocv.lm 0; hdlt 1)
template <int N>
static bool solve_gauss_n(const double* Ain, const double* bin, double* x) {
    double A[N][N + 1];
    for (int i = 0; i < N; ++i) { for (int j = 0; j < N; ++j) A[i][j] = Ain[i * N + j]; A[i][N] = bin[i]; }
    for (int c = 0; c < N; ++c) {
        int p = c;
        for (int r = c + 1; r < N; ++r) if (std::fabs(A[r][c]) > std::fabs(A[p][c])) p = r;
        if (A[p][c] == 0.0) return false;
        if (p != c) for (int j = 0; j <= N; ++j) std::swap(A[p][j], A[c][j]);
        for (int r = c + 1; r < N; ++r) {
            double f = A[r][c] / A[c][c];
            for (int j = c; j <= N; ++j) A[r][j] -= f * A[c][j];
        }
    }
    for (int i = N - 1; i >= 0; --i) {
        double sm = A[i][N];
        for (int j = i + 1; j < N; ++j) sm -= A[i][j] * x[j];
        x[i] = sm / A[i][i];
    }
    return true;
}

// ocv.hdlt 1 (definitional cross-check, minimal samples only): the same normalisation, then the 8 equations
// of 4 pairs with h33 = 1 solved directly.
static int homography_4pt_direct(const P2f* M, const P2f* m, double H[9]) {
    const int count = 4;
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; ++i) { cmx += m[i].x; cmy += m[i].y; cMx += M[i].x; cMy += M[i].y; }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; ++i) {
        smx += std::fabs(m[i].x - cmx); smy += std::fabs(m[i].y - cmy);
        sMx += std::fabs(M[i].x - cMx); sMy += std::fabs(M[i].y - cMy);
    }
    if (std::fabs(smx) < DBL_EPSILON || std::fabs(smy) < DBL_EPSILON || std::fabs(sMx) < DBL_EPSILON || std::fabs(sMy) < DBL_EPSILON) return 0;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    double A[64], b[8], h[9];
    for (int i = 0; i < 4; ++i) {
        double x = (m[i].x - cmx) * smx, y = (m[i].y - cmy) * smy;
        double X = (M[i].x - cMx) * sMx, Y = (M[i].y - cMy) * sMy;
        double r0[8] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y}, r1[8] = {0, 0, 0, X, Y, 1, -y * X, -y * Y};
        std::memcpy(A + 16 * i, r0, sizeof(r0)); std::memcpy(A + 16 * i + 8, r1, sizeof(r1));
        b[2 * i] = x; b[2 * i + 1] = y;
    }
    if (!solve_gauss_n<8>(A, b, h)) return 0;
    h[8] = 1;
    const double ix = 1. / smx, iy = 1. / smy;
    double T[9];
    for (int j = 0; j < 3; ++j) { T[j] = ix * h[j] + cmx * h[6 + j]; T[3 + j] = iy * h[3 + j] + cmy * h[6 + j]; T[6 + j] = h[6 + j]; }
    const double n2 = -cMx * sMx, n5 = -cMy * sMy;
    double R[9];
    for (int r = 0; r < 3; ++r) { R[3 * r] = T[3 * r] * sMx; R[3 * r + 1] = T[3 * r + 1] * sMy; R[3 * r + 2] = T[3 * r] * n2 + T[3 * r + 1] * n5 + T[3 * r + 2]; }
    const double sc = 1. / R[8];
    for (int j = 0; j < 9; ++j) H[j] = R[j] * sc;
    return 1;
}

// ocv.hdlt 2: the same normalisation, then the 4-point model in CLOSED FORM — the projective map of the unit square onto each
// normalised quadrilateral (Heckbert 1989), H = S_to * adj(S_from): ~90 multiplications, 2 divisions, no pivoting and no
// iteration.  Exact for 4 pairs like the other two forms (equal to f64 round-off); every product and sum below is evaluated in
// the order written (the HIP twin, csrc/homography.hip.h dlt4_closed, is this code).  A quadrilateral whose points 1, 2, 3 are
// collinear has no such map: no model (checkSubset has removed those samples already).
static bool square_to_quad(const double* x, const double* y, double* S) {
    const double dx1 = x[1] - x[2], dx2 = x[3] - x[2], dx3 = x[0] - x[1] + x[2] - x[3];
    const double dy1 = y[1] - y[2], dy2 = y[3] - y[2], dy3 = y[0] - y[1] + y[2] - y[3];
    const double det = dx1 * dy2 - dx2 * dy1;
    if (det == 0.0) return false;
    const double g = (dx3 * dy2 - dx2 * dy3) / det, hh = (dx1 * dy3 - dx3 * dy1) / det;
    S[0] = x[1] - x[0] + g * x[1]; S[1] = x[3] - x[0] + hh * x[3]; S[2] = x[0];
    S[3] = y[1] - y[0] + g * y[1]; S[4] = y[3] - y[0] + hh * y[3]; S[5] = y[0];
    S[6] = g; S[7] = hh; S[8] = 1.0;
    return true;
}
static int homography_4pt_closed(const P2f* M, const P2f* m, double H[9]) {
    const int count = 4;
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; ++i) { cmx += m[i].x; cmy += m[i].y; cMx += M[i].x; cMy += M[i].y; }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; ++i) {
        smx += std::fabs(m[i].x - cmx); smy += std::fabs(m[i].y - cmy);
        sMx += std::fabs(M[i].x - cMx); sMy += std::fabs(M[i].y - cMy);
    }
    if (std::fabs(smx) < DBL_EPSILON || std::fabs(smy) < DBL_EPSILON || std::fabs(sMx) < DBL_EPSILON || std::fabs(sMy) < DBL_EPSILON) return 0;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    double fx[4], fy[4], tx[4], ty[4];
    for (int i = 0; i < 4; ++i) {
        tx[i] = (m[i].x - cmx) * smx; ty[i] = (m[i].y - cmy) * smy;
        fx[i] = (M[i].x - cMx) * sMx; fy[i] = (M[i].y - cMy) * sMy;
    }
    double A[9], B[9];
    if (!square_to_quad(fx, fy, A) || !square_to_quad(tx, ty, B)) return 0;
    // adj(A)
    const double J[9] = {A[4] * A[8] - A[5] * A[7], A[2] * A[7] - A[1] * A[8], A[1] * A[5] - A[2] * A[4],
                         A[5] * A[6] - A[3] * A[8], A[0] * A[8] - A[2] * A[6], A[2] * A[3] - A[0] * A[5],
                         A[3] * A[7] - A[4] * A[6], A[1] * A[6] - A[0] * A[7], A[0] * A[4] - A[1] * A[3]};
    double h[9];
    for (int r = 0; r < 3; ++r)
        for (int c2 = 0; c2 < 3; ++c2) h[3 * r + c2] = B[3 * r] * J[c2] + B[3 * r + 1] * J[3 + c2] + B[3 * r + 2] * J[6 + c2];
    const double ix = 1. / smx, iy = 1. / smy;
    double T[9];
    for (int j = 0; j < 3; ++j) { T[j] = ix * h[j] + cmx * h[6 + j]; T[3 + j] = iy * h[3 + j] + cmy * h[6 + j]; T[6 + j] = h[6 + j]; }
    const double n2 = -cMx * sMx, n5 = -cMy * sMy;
    double R[9];
    for (int r = 0; r < 3; ++r) { R[3 * r] = T[3 * r] * sMx; R[3 * r + 1] = T[3 * r + 1] * sMy; R[3 * r + 2] = T[3 * r] * n2 + T[3 * r + 1] * n5 + T[3 * r + 2]; }
    const double sc = 1. / R[8];
    for (int j = 0; j < 9; ++j) H[j] = R[j] * sc;
    return 1;
}

// precomp.hpp haveCollinearPoints: only the LAST point of the subset is tested against the pairs before it
static bool have_collinear_points(const P2f* p, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        double dx1 = p[j].x - p[i].x, dy1 = p[j].y - p[i].y;        // f32 differences widened, as `double dx1 = ptr[j].x - ptr[i].x`
        for (int k = 0; k < j; ++k) {
            double dx2 = p[k].x - p[i].x, dy2 = p[k].y - p[i].y;
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2))) return true;
        }
    }
    return false;
}

static inline double det3_rows(const P2f& a, const P2f& b, const P2f& c) {   // Matx33d(a.x, a.y, 1, b.x, b.y, 1, c.x, c.y, 1)
    const double a00 = a.x, a01 = a.y, a02 = 1., a10 = b.x, a11 = b.y, a12 = 1., a20 = c.x, a21 = c.y, a22 = 1.;
    return a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
}

// HomographyEstimatorCallback::checkSubset (count == 4): no collinear / coincident triple through the last point in
// either set, and the four triples keep or all flip their orientation (Marquez-Neila et al. 2013)
static bool homography_check_subset(const P2f* ms1, const P2f* ms2, int count) {
    if (have_collinear_points(ms1, count) || have_collinear_points(ms2, count)) return false;
    if (count == 4) {
        static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (int i = 0; i < 4; ++i) {
            const int* t = tt[i];
            negative += det3_rows(ms1[t[0]], ms1[t[1]], ms1[t[2]]) * det3_rows(ms2[t[0]], ms2[t[1]], ms2[t[2]]) < 0;
        }
        if (negative != 0 && negative != 4) return false;
    }
    return true;
}

// HomographyEstimatorCallback::computeError + findInliers: f32 re-projection error against (float)(thr * thr)
static int homography_find_inliers(const P2f* M, const P2f* m, int n, const double H[9], float thr2, uint8_t* mask) {
    const float Hf[8] = {(float)H[0], (float)H[1], (float)H[2], (float)H[3], (float)H[4], (float)H[5], (float)H[6], (float)H[7]};
    int good = 0;
    for (int i = 0; i < n; ++i) {
        float ww = 1.f / (Hf[6] * M[i].x + Hf[7] * M[i].y + 1.f);
        float dx = (Hf[0] * M[i].x + Hf[1] * M[i].y + Hf[2]) * ww - m[i].x;
        float dy = (Hf[3] * M[i].x + Hf[4] * M[i].y + Hf[5]) * ww - m[i].y;
        float e = dx * dx + dy * dy;
        int f = e <= thr2;
        mask[i] = (uint8_t)f;
        good += f;
    }
    return good;
}

// HomographyRefineCallback::compute: residuals and (optionally) J^T J, J^T r for h = H[0..7]; returns |r|^2
static double lm8_eval(const P2f* M, const P2f* m, int n, const double h[8], double* A, double* v, double* rinf) {
    double S = 0, ri = 0;
    if (A) { std::fill(A, A + 64, 0.0); std::fill(v, v + 8, 0.0); }
    for (int i = 0; i < n; ++i) {
        double Mx = M[i].x, My = M[i].y;
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = std::fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        double ex = xi - m[i].x, ey = yi - m[i].y;
        S += ex * ex; S += ey * ey;
        ri = std::max(ri, std::max(std::fabs(ex), std::fabs(ey)));
        if (A) {
            const double J0[8] = {Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi};
            const double J1[8] = {0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi};
            for (int a = 0; a < 8; ++a) {
                for (int b = 0; b < 8; ++b) A[a * 8 + b] += J0[a] * J0[b] + J1[a] * J1[b];
                v[a] += J0[a] * ex + J1[a] * ey;
            }
        }
    }
    if (rinf) *rinf = ri;
    return S;
}

// calib3d/src/levmarq.cpp LMSolverImpl::run on the 8 parameters (the same loop as lm_refine above)
static void lm8_refine(const P2f* M, const P2f* m, int n, double h[8], int max_iters, int lm_variant) {
    const double eps = (double)FLT_EPSILON;
    double x[8], xd[8], A[64], v[8], D[8], d[8], Ap[64], rinf = 0;
    std::memcpy(x, h, sizeof(x));
    auto solve = [&](const double* a, const double* b, double* o) { return lm_variant == 1 ? solve_eig_n<8>(a, b, o) : solve_gauss_n<8>(a, b, o); };
    double S = lm8_eval(M, m, n, x, A, v, &rinf);
    for (int i = 0; i < 8; ++i) D[i] = A[i * 8 + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        std::memcpy(Ap, A, sizeof(A));
        for (int i = 0; i < 8; ++i) Ap[i * 8 + i] += lambda * D[i];
        if (!solve(Ap, v, d)) std::fill(d, d + 8, 0.0);
        for (int i = 0; i < 8; ++i) xd[i] = x[i] - d[i];
        double Sd = lm8_eval(M, m, n, xd, nullptr, nullptr, nullptr);
        double dS = 0;
        for (int i = 0; i < 8; ++i) {
            double t = 2 * v[i];
            for (int j = 0; j < 8; ++j) t -= A[i * 8 + j] * d[j];
            dS += d[i] * t;
        }
        double R = (S - Sd) / (std::fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double t = 0;
            for (int i = 0; i < 8; ++i) t += d[i] * v[i];
            double nu = (Sd - S) / (std::fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = std::min(std::max(nu, 2.), 10.);
            if (lambda == 0) {
                double maxval = DBL_EPSILON;
                for (int i = 0; i < 8; ++i) {
                    double e[8] = {0, 0, 0, 0, 0, 0, 0, 0}, col[8];
                    e[i] = 1;
                    if (solve(A, e, col)) maxval = std::max(maxval, std::fabs(col[i]));
                }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            std::memcpy(x, xd, sizeof(x));
            lm8_eval(M, m, n, x, A, v, &rinf);
        }
        iter++;
        double dinf = 0;
        for (int i = 0; i < 8; ++i) dinf = std::max(dinf, std::fabs(d[i]));
        if (!(iter < max_iters && dinf >= eps && rinf >= eps)) break;
    }
    std::memcpy(h, x, sizeof(x));
}

struct HomographyStats { int iters = 0, attempts = 0, rotations = 0, draws = 0; };

// findHomography(RANSAC) = RANSACPointSetRegistrator::run with modelPoints 4 + the refinement of fundam.cpp.
static bool find_homography(const P2f* from, const P2f* to, int count, const slideo_config& c, double H[9], uint8_t* mask,
                            HomographyStats* st = nullptr) {
    const int model_points = 4;
    std::fill(H, H + 9, 0.0);
    std::fill(mask, mask + count, (uint8_t)0);
    if (count < model_points) return false;
    bool result = false;
    if (count == model_points) {                                        // `method == 0 || npoints == 4`: the kernel alone
        result = (c.ocv.hdlt == 2 ? homography_4pt_closed(from, to, H) : c.ocv.hdlt == 1 ? homography_4pt_direct(from, to, H) : homography_dlt(from, to, count, H)) > 0;
        if (result) std::fill(mask, mask + count, (uint8_t)1);
        else std::fill(H, H + 9, 0.0);
        return result;
    }
    const float thr2 = (float)(c.ransac_threshold * c.ransac_threshold);
    int niters = std::max(c.ransac_max_iters, 1), max_good = 0, iter;
    CvRng rng((uint64_t)-1, c.ocv.rng_mul);
    std::vector<uint8_t> cur(count);
    double Hi[9];
    for (iter = 0; iter < niters; ++iter) {
        // getSubset(m1, m2, ms1, ms2, rng, 10000): 4 distinct indices (a duplicate is redrawn), the whole subset
        // redrawn while checkSubset rejects it, at most 10000 attempts
        P2f f[4], t[4];
        bool found = false;
        for (int attempt = 0; attempt < 10000; ++attempt) {
            int idx[4];
            for (int i = 0; i < model_points; ++i) {
                int idx_i;
                for (idx_i = rng.uniform(0, count); std::find(idx, idx + i, idx_i) != idx + i; idx_i = rng.uniform(0, count)) { if (st) st->draws++; }
                if (st) st->draws++;
                idx[i] = idx_i;
                f[i] = from[idx_i]; t[i] = to[idx_i];
            }
            if (st) st->attempts++;
            if (homography_check_subset(f, t, model_points)) { found = true; break; }
        }
        if (!found) {
            if (iter == 0) { std::fill(H, H + 9, 0.0); return false; }
            break;
        }
        int rot = 0;
        int nmodels = c.ocv.hdlt == 2 ? homography_4pt_closed(f, t, Hi) : c.ocv.hdlt == 1 ? homography_4pt_direct(f, t, Hi) : homography_dlt(f, t, model_points, Hi, &rot);
        if (st) st->rotations += rot;
        if (nmodels <= 0) continue;
        int good = homography_find_inliers(from, to, count, Hi, thr2, cur.data());
        if (good > std::max(max_good, model_points - 1)) {
            std::memcpy(mask, cur.data(), count);
            std::memcpy(H, Hi, sizeof(Hi));
            max_good = good;
            niters = ransac_update_iters(c.ransac_confidence, (double)(count - good) / count, model_points, niters);
        }
    }
    if (st) st->iters = iter;
    result = max_good > 0;
    if (!result) { std::fill(H, H + 9, 0.0); std::fill(mask, mask + count, (uint8_t)0); return false; }
    if (c.refine_iters > 0) {                                           // `result && npoints > 4`: DLT over the inliers, then LM
        std::vector<P2f> s, d;
        for (int i = 0; i < count; ++i) if (mask[i]) { s.push_back(from[i]); d.push_back(to[i]); }
        if (!s.empty()) {
            homography_dlt(s.data(), d.data(), (int)s.size(), H);       // (return value ignored, as fundam.cpp does)
            lm8_refine(s.data(), d.data(), (int)s.size(), H, c.refine_iters, c.ocv.lm);
        }
    }
    return true;
}

// ---------------------------------------------------------------------------
// [OCV A.10] warpAffine, nearest, WARP_INVERSE_MAP, BORDER_CONSTANT(0)
// imgproc/src/imgwarp.cpp (mo/lib.rs:338-348).  M maps dst (slide) -> src (frame).
// ---------------------------------------------------------------------------
static inline int sat_int(double v) {
    if (!(v > -2147483648.0)) return INT32_MIN;   // also NaN
    if (v >= 2147483647.0) return INT32_MAX;
    return cv_round(v);
}
static inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

struct WarpSampler {  // source coordinate of destination pixel (x, y)
    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1};
    int variant = 0;      // ocv.warp: 0 = imgwarp.cpp's 10-bit fixed point, 1 = cvRound of the f64 coordinate
    int persp = 0;        // verify_model 1: warpPerspective (M is 3x3), else warpAffine (M[0..5])
    int bw0 = 64;         // warpPerspective walks the destination in blocks of bw0 columns (set_dst_size)
    // WarpPerspectiveInvoker: BLOCK_SZ 32, bh0 = min(16, h), bw0 = min(1024 / bh0, w) — the block origin enters the
    // floating-point association of the coordinates
    void set_dst_size(int dw, int dh) { int bh0 = std::min(16, dh); bw0 = std::max(1, std::min(32 * 32 / std::max(bh0, 1), dw)); }
    inline void src_xy(int x, int y, int& sx, int& sy) const {
        if (persp) {
            // imgproc/src/imgwarp.cpp WarpPerspectiveInvoker, INTER_NEAREST (recalled): per destination row of a block
            // X0 = M0 x_blk + M1 y + M2 ..., per pixel W = W0 + M6 x1; W = W ? 1/W : 0; fX = clamp((X0 + M0 x1) W)
            const int xb = (x / bw0) * bw0, x1 = x - xb;
            const double X0 = M[0] * xb + M[1] * y + M[2], Y0 = M[3] * xb + M[4] * y + M[5], W0 = M[6] * xb + M[7] * y + M[8];
            double W = W0 + M[6] * x1;
            W = W ? 1. / W : 0;
            const double fX = std::max((double)INT32_MIN, std::min((double)INT32_MAX, (X0 + M[0] * x1) * W));
            const double fY = std::max((double)INT32_MIN, std::min((double)INT32_MAX, (Y0 + M[3] * x1) * W));
            sx = sat_short(sat_int(fX)); sy = sat_short(sat_int(fY));
            return;
        }
        if (variant == 1) {
            sx = sat_short(sat_int(M[0] * x + M[1] * y + M[2])); sy = sat_short(sat_int(M[3] * x + M[4] * y + M[5]));
            return;
        }
        const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, round_delta = AB_SCALE / 2;
        int adelta = sat_int(M[0] * x * AB_SCALE), bdelta = sat_int(M[3] * x * AB_SCALE);
        int X0 = sat_int((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        int Y0 = sat_int((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        // two's-complement wrap-around add, as the int arithmetic in imgwarp.cpp
        int X = (int)((uint32_t)X0 + (uint32_t)adelta) >> AB_BITS;
        int Y = (int)((uint32_t)Y0 + (uint32_t)bdelta) >> AB_BITS;
        sx = sat_short(X); sy = sat_short(Y);
    }
};

static void warp_affine_nn_bgr8(const uint8_t* src, int sw, int sh, int sstride, const double M[6],
                                uint8_t* dst, int dw, int dh, int variant = 0) {
    WarpSampler ws;
    std::memcpy(ws.M, M, 6 * sizeof(double));
    ws.variant = variant;
    for (int y = 0; y < dh; ++y) {
        uint8_t* d = dst + (size_t)y * dw * 3;
        for (int x = 0; x < dw; ++x) {
            int sx, sy;
            ws.src_xy(x, y, sx, sy);
            if ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) {
                const uint8_t* s = src + (size_t)sy * sstride + 3 * sx;
                d[3 * x] = s[0]; d[3 * x + 1] = s[1]; d[3 * x + 2] = s[2];
            } else {
                d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = 0;
            }
        }
    }
}

// warpPerspective(src, H, dsize, WARP_INVERSE_MAP [nearest], BORDER_CONSTANT 0): H maps dst -> src
static void warp_perspective_nn_bgr8(const uint8_t* src, int sw, int sh, int sstride, const double H[9],
                                     uint8_t* dst, int dw, int dh) {
    WarpSampler ws;
    std::memcpy(ws.M, H, 9 * sizeof(double));
    ws.persp = 1; ws.set_dst_size(dw, dh);
    for (int y = 0; y < dh; ++y) {
        uint8_t* d = dst + (size_t)y * dw * 3;
        for (int x = 0; x < dw; ++x) {
            int sx, sy;
            ws.src_xy(x, y, sx, sy);
            if ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) {
                const uint8_t* sp = src + (size_t)sy * sstride + 3 * sx;
                d[3 * x] = sp[0]; d[3 * x + 1] = sp[1]; d[3 * x + 2] = sp[2];
            } else {
                d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = 0;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// [OCV A.11] resize INTER_AREA (shrink), imgproc/src/resize.cpp
// computeResizeAreaTab + ResizeArea_Invoker (f32 accumulators), with the
// integer-scale fast paths (ResizeAreaFast_Invoker, 2x2 special case).
// ---------------------------------------------------------------------------
struct AreaTap { int si, di; float alpha; };

// ocv.area: 0 = computeResizeAreaTab as below, 1 = exact box-overlap weights (no 1e-3 cut-off, weights normalised by the
// clipped cell width) — a definitional cross-check
static void area_tab(int ssize, int dsize, double scale, std::vector<AreaTap>& tab, int variant = 0) {
    tab.clear();
    if (variant == 1) {
        for (int dx = 0; dx < dsize; ++dx) {
            double a = dx * scale, b = std::min(a + scale, (double)ssize);
            double cell = b - a;
            for (int sx = cv_floor(a); sx < ssize && (double)sx < b; ++sx) {
                double ov = std::min(b, sx + 1.0) - std::max(a, (double)sx);
                if (ov > 0) tab.push_back({sx, dx, (float)(ov / cell)});
            }
        }
        return;
    }
    for (int dx = 0; dx < dsize; ++dx) {
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cell = std::min(scale, ssize - fsx1);
        int sx1 = cv_ceil(fsx1), sx2 = cv_floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) tab.push_back({sx1 - 1, dx, (float)((sx1 - fsx1) / cell)});
        for (int sx = sx1; sx < sx2; ++sx) tab.push_back({sx, dx, (float)(1.0 / cell)});
        if (fsx2 - sx2 > 1e-3)
            tab.push_back({sx2, dx, (float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)});
    }
}

static inline uint8_t sat_u8_f(float v) {
    int i = (int)std::lrintf(v);
    return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

// Generic over a pixel fetcher so the same arithmetic serves plain images and
// the fused warp -> area path used by tests of the fused GPU kernel.
template <class Fetch>
static bool resize_area_generic(int sw, int sh, int dw, int dh, uint8_t* dst, Fetch fetch, int area_variant = 0) {
    if (dw <= 0 || dh <= 0 || dw > sw || dh > sh) return false;   // shrink only
    double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    int iscale_x = (int)std::min<long long>(std::llrint(scale_x), INT32_MAX);  // saturate_cast<int>
    int iscale_y = (int)std::min<long long>(std::llrint(scale_y), INT32_MAX);
    bool fast = std::fabs(scale_x - iscale_x) < DBL_EPSILON && std::fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (fast) {
        int area = iscale_x * iscale_y;
        float scale = 1.f / (float)area;
        // ResizeAreaFast: only complete cells are averaged; dst cells beyond
        // (src - iscale)/iscale fall back to the clipped sum (never hit for exact multiples,
        // which is the only way dsize*iscale <= ssize here).
        for (int dy = 0; dy < dh; ++dy)
            for (int dx = 0; dx < dw; ++dx) {
                int sum[3] = {0, 0, 0};
                for (int yy = 0; yy < iscale_y; ++yy)
                    for (int xx = 0; xx < iscale_x; ++xx) {
                        uint8_t p[3];
                        int sx = std::min(dx * iscale_x + xx, sw - 1), sy = std::min(dy * iscale_y + yy, sh - 1);
                        fetch(sx, sy, p);
                        sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2];
                    }
                for (int ch = 0; ch < 3; ++ch) {
                    uint8_t o;
                    if (iscale_x == 2 && iscale_y == 2) o = (uint8_t)((sum[ch] + 2) >> 2);
                    else o = sat_u8_f((float)sum[ch] * scale);
                    dst[((size_t)dy * dw + dx) * 3 + ch] = o;
                }
            }
        return true;
    }
    std::vector<AreaTap> xt, yt;
    area_tab(sw, dw, scale_x, xt, area_variant);
    area_tab(sh, dh, scale_y, yt, area_variant);
    // group taps by destination index
    std::vector<int> xs(dw + 1, 0), ys(dh + 1, 0);
    for (const AreaTap& t : xt) xs[t.di + 1]++;
    for (const AreaTap& t : yt) ys[t.di + 1]++;
    for (int i = 0; i < dw; ++i) xs[i + 1] += xs[i];
    for (int i = 0; i < dh; ++i) ys[i + 1] += ys[i];
    for (int dy = 0; dy < dh; ++dy)
        for (int dx = 0; dx < dw; ++dx) {
            float sum[3] = {0, 0, 0};
            for (int j = ys[dy]; j < ys[dy + 1]; ++j) {
                float beta = yt[j].alpha;
                float buf[3] = {0, 0, 0};
                for (int k = xs[dx]; k < xs[dx + 1]; ++k) {
                    uint8_t p[3];
                    fetch(xt[k].si, yt[j].si, p);
                    float al = xt[k].alpha;
                    buf[0] = buf[0] + (float)p[0] * al;
                    buf[1] = buf[1] + (float)p[1] * al;
                    buf[2] = buf[2] + (float)p[2] * al;
                }
                if (j == ys[dy]) { sum[0] = beta * buf[0]; sum[1] = beta * buf[1]; sum[2] = beta * buf[2]; }
                else { sum[0] += beta * buf[0]; sum[1] += beta * buf[1]; sum[2] += beta * buf[2]; }
            }
            for (int ch = 0; ch < 3; ++ch) dst[((size_t)dy * dw + dx) * 3 + ch] = sat_u8_f(sum[ch]);
        }
    return true;
}

// mo/image_utils.rs:8-20 to_small_image: target size
static void small_size(int w, int h, int small_area, int& sw, int& sh) {
    float factor = std::sqrt((float)small_area / (float)(w * h));
    sw = (int)((float)w * factor);
    sh = (int)((float)h * factor);
}

static bool small_image(const uint8_t* bgr, int w, int h, int stride, int small_area,
                        std::vector<uint8_t>& out, int& sw, int& sh, int area_variant = 0) {
    small_size(w, h, small_area, sw, sh);
    out.assign((size_t)std::max(sw, 0) * std::max(sh, 0) * 3, 0);
    return resize_area_generic(w, h, sw, sh, out.data(), [&](int x, int y, uint8_t* p) {
        const uint8_t* s = bgr + (size_t)y * stride + 3 * x;
        p[0] = s[0]; p[1] = s[1]; p[2] = s[2];
    }, area_variant);
}

// mo/image_utils.rs:22-27 compute_similarity  ([OCV A.12] norm L2: integer sum, f64 sqrt)
static float similarity_bgr8(const uint8_t* a, const uint8_t* b, int w, int h) {
    uint64_t ss = 0;
    size_t n = (size_t)w * h * 3;
    for (size_t i = 0; i < n; ++i) { int d = (int)a[i] - (int)b[i]; ss += (uint64_t)(d * d); }
    double err = std::sqrt((double)ss);
    int p = w * h;
    float max_error = std::sqrt((255.0f * 255.0f * 3.0f) * (float)p);
    return 1.0f - (float)err / max_error;
}

// ---------------------------------------------------------------------------
// Page DB + per-frame decision (mo/lib.rs:92-131, 249-413)
// ---------------------------------------------------------------------------
struct Page {
    int w, h;
    OrbResult orb;                    // (SIFT mode: kp = the SIFT keypoints, desc unused)
    std::vector<uint8_t> sdesc;       // SIFT mode: 128 bytes per keypoint
    std::vector<uint8_t> small;
    int sw, sh;
};

}  // namespace

namespace {
// slideo_config.matcher 1 — the candidate rule of FLANN's LshIndex as the reference configures it (mo/flann.rs:14-26:
// table_number 6, key_size 12, multi_probe_level 1), flann/lsh_index.h + lsh_table.h RECALLED: table i hashes a descriptor by
// key_size of its 256 bits (cv::randShuffle of the bit positions on the thread's default cv::RNG, first key_size of the
// shuffled array; keys pack the bits in ascending position order); a query probes, in every table, the buckets whose key is
// within multi_probe_level bits of its own; the candidates are scored exactly and the k best returned.  Departure: FLANN's
// KNNUniqueResultSet rejects a candidate whose distance EQUALS the current k-th, so ties at the k-th place depend on the
// visiting order; here the result is the k smallest (distance, row) of the candidate set (canonical).
struct LshIdx {
    int ntab = 0, kb = 0, mp = 0;
    std::vector<int> bit;                         // [ntab][kb] ascending
    std::vector<std::vector<int32_t>> ofs, rows;  // per table: bucket offsets (2^kb + 1), rows grouped by key (ascending inside a bucket)
    uint32_t key(int t, const uint8_t* d) const {
        uint32_t k = 0;
        for (int b = 0; b < kb; ++b) { const int p = bit[(size_t)t * kb + b]; k |= (uint32_t)((d[p >> 3] >> (p & 7)) & 1) << b; }
        return k;
    }
    void build(const slideo_config& c, const uint8_t* train, int M) {
        ntab = c.lsh_tables; kb = c.lsh_key_bits; mp = c.lsh_multi_probe;
        bit.assign((size_t)ntab * kb, 0);
        CvRng rng(0xffffffffULL, c.ocv.rng_mul);          // cv::theRNG()'s initial state
        for (int t = 0; t < ntab; ++t) {
            int a[256];
            for (int i = 0; i < 256; ++i) a[i] = i;
            for (int i = 0; i < 256; ++i) { const int j = (int)(rng.next() % 256u); std::swap(a[j], a[i]); }     // cv::randShuffle_<int>
            std::sort(a, a + kb);
            for (int b = 0; b < kb; ++b) bit[(size_t)t * kb + b] = a[b];
        }
        ofs.assign(ntab, std::vector<int32_t>()); rows.assign(ntab, std::vector<int32_t>());
        for (int t = 0; t < ntab; ++t) {
            std::vector<int32_t>& o = ofs[t];
            o.assign(((size_t)1 << kb) + 1, 0);
            std::vector<uint32_t> k(M);
            for (int i = 0; i < M; ++i) { k[i] = key(t, train + (size_t)i * 32); o[k[i] + 1]++; }
            for (size_t i = 0; i + 1 < o.size(); ++i) o[i + 1] += o[i];
            std::vector<int32_t> cur(o.begin(), o.end() - 1);
            rows[t].assign(M, 0);
            for (int i = 0; i < M; ++i) rows[t][cur[k[i]]++] = i;
        }
    }
    void knn(const uint8_t* q, const uint8_t* train, int k, int32_t* idx, uint16_t* dist) const {
        std::vector<uint32_t> cand;
        for (int t = 0; t < ntab; ++t) {
            const uint32_t qk = key(t, q);
            for (uint32_t m = 0; m < (1u << kb); ++m) {
                if (__builtin_popcount(m) > mp) continue;
                const uint32_t b = qk ^ m;
                for (int32_t j = ofs[t][b]; j < ofs[t][b + 1]; ++j) {
                    const int32_t row = rows[t][j];
                    cand.push_back(((uint32_t)hamming256(q, train + (size_t)row * 32) << 23) | (uint32_t)row);
                }
            }
        }
        std::sort(cand.begin(), cand.end());
        cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
        for (int r = 0; r < k; ++r) {
            idx[r] = r < (int)cand.size() ? (int32_t)(cand[r] & 0x7FFFFFu) : -1;
            dist[r] = r < (int)cand.size() ? (uint16_t)(cand[r] >> 23) : (uint16_t)65535;
        }
    }
};
}  // namespace

#include "sift_oracle.h"

struct so_pagedb {
    slideo_config cfg;
    // so_pagedb_use_sift (the product's slideo_matcher_use_sift): SIFT features, squared-L2 2-NN + Lowe's ratio test in front of
    // the path's own vote / RANSAC / re-projection stages
    bool sift = false;
    so_sift_config sc{};
    float ratio = 0.75f;
    std::vector<uint8_t> train128;    // M x 128 (SIFT mode)
    LshIdx lsh;
    std::vector<Page> pages;
    std::vector<uint8_t> train;       // M x 32
    std::vector<uint64_t> train_blocked;   // the same rows in knn_hamming_blocked's layout
    std::vector<int32_t> train_page;  // M
    std::vector<int32_t> page_ofs;    // P+1
    bool finalized = false;
};

extern "C" void so_knn_l2_u8(const uint8_t* q, int nq, const uint8_t* t, int nt, int k, int32_t* idx, uint32_t* dist);

namespace {

struct Vote { int q, t_local; };

static void match_frame(const so_pagedb& db, const uint8_t* bgr, int w, int h, int stride,
                        slideo_verdict& out, std::vector<slideo_candidate>* trace) {
    const slideo_config& c = db.cfg;
    out.page_idx = -1; out.similarity = 0; out.inliers = 0; out.n_keypoints = 0;
    if (trace) trace->clear();
    OrbResult fr;
    int P = (int)db.pages.size();
    std::vector<std::vector<Vote>> votes(P);
    if (db.sift) {
        // SIFT mode (no reference counterpart): cv::SIFT features, BFMatcher(NORM_L2).knnMatch(k = 2), Lowe's ratio test on the
        // distances as BFMatcher returns them (f32 square roots); a query with fewer than two neighbours casts no vote
        SiftResult sr;
        sift_detect_describe(bgr, w, h, stride, db.sc, c.ocv, sr);
        fr.kp = sr.kp;
        const int K = (int)fr.kp.size(), M = (int)db.train_page.size();
        out.n_keypoints = K;
        if (K == 0) return;
        const int kq = db.ratio > 0.f ? 2 : c.knn_k;
        std::vector<int32_t> idx((size_t)K * kq);
        std::vector<uint32_t> d2((size_t)K * kq);
        so_knn_l2_u8(sr.desc.data(), K, db.train128.data(), M, kq, idx.data(), d2.data());
        for (int q = 0; q < K; ++q) {
            if (db.ratio > 0.f) {
                if (idx[(size_t)q * 2] < 0 || idx[(size_t)q * 2 + 1] < 0) continue;
                const float a = std::sqrt((float)d2[(size_t)q * 2]), b = std::sqrt((float)d2[(size_t)q * 2 + 1]);
                if (a < db.ratio * b) {
                    const int ti = idx[(size_t)q * 2], pg = db.train_page[ti];
                    votes[pg].push_back({q, ti - db.page_ofs[pg]});
                }
                continue;
            }
            // ratio 0: the path's tolerance vote (mo/lib.rs:268-282) on the L2 distances
            if (idx[(size_t)q * kq] < 0) continue;
            const float lim = std::sqrt((float)d2[(size_t)q * kq]) * c.vote_tolerance;
            for (int r = 0; r < kq; ++r) {
                const int ti = idx[(size_t)q * kq + r];
                if (ti < 0) break;
                if (std::sqrt((float)d2[(size_t)q * kq + r]) < lim) {
                    const int pg = db.train_page[ti];
                    votes[pg].push_back({q, ti - db.page_ofs[pg]});
                }
            }
        }
    } else {
    orb_detect_describe(bgr, w, h, stride, c, fr);           // mo/lib.rs:264-265
    int K = (int)fr.kp.size(), M = (int)db.train_page.size(), k = c.knn_k;
    out.n_keypoints = K;
    if (K == 0) return;
    std::vector<int32_t> idx((size_t)K * k);
    std::vector<uint16_t> dist((size_t)K * k);
    if (c.matcher == 1)                                       // the reference's index: LSH candidates only
        for (int qi = 0; qi < K; ++qi) db.lsh.knn(fr.desc.data() + (size_t)qi * 32, db.train.data(), k, idx.data() + (size_t)qi * k, dist.data() + (size_t)qi * k);
    else
        knn_hamming_blocked(fr.desc.data(), K, db.train_blocked.data(), M, k, idx.data(), dist.data());  // :266 (== knn_hamming)
    // tolerance vote, mo/lib.rs:268-282: d < best * 1.05 (f32, strict)
    for (int q = 0; q < K; ++q) {
        if (idx[(size_t)q * k] < 0) continue;
        float best = (float)dist[(size_t)q * k];
        if (c.ratio_test > 0.f) {
            // extension (no reference counterpart; include/slideo_amd.h ratio_test): the ratio test on the two
            // nearest rows replaces the tolerance vote
            if (k >= 2 && idx[(size_t)q * k + 1] >= 0 && best < c.ratio_test * (float)dist[(size_t)q * k + 1]) {
                int ti = idx[(size_t)q * k];
                int pg = db.train_page[ti];
                votes[pg].push_back({q, ti - db.page_ofs[pg]});
            }
            continue;
        }
        float lim = best * c.vote_tolerance;
        for (int r = 0; r < k; ++r) {
            int ti = idx[(size_t)q * k + r];
            if (ti < 0) break;
            if ((float)dist[(size_t)q * k + r] < lim) {
                int pg = db.train_page[ti];
                votes[pg].push_back({q, ti - db.page_ofs[pg]});
            }
        }
    }
    }   // ORB / Hamming front end
    // candidate ranking, mo/lib.rs:284-295: stable by count desc over ascending page index
    std::vector<int> cand;
    for (int p = 0; p < P; ++p) if (!votes[p].empty()) cand.push_back(p);
    std::stable_sort(cand.begin(), cand.end(),
                     [&](int a, int b) { return votes[a].size() > votes[b].size(); });
    if ((int)cand.size() > c.max_candidate_pages) cand.resize(c.max_candidate_pages);

    struct Rated { int page; int nvotes; double rating; double M[9]; bool found; float sim; bool survived; };
    std::vector<Rated> rated;
    for (int p : cand) {                                       // mo/lib.rs:296-313
        const std::vector<Vote>& v = votes[p];
        std::vector<P2f> from(v.size()), to(v.size());
        for (size_t i = 0; i < v.size(); ++i) {
            const slideo_keypoint& sk = db.pages[p].orb.kp[v[i].t_local];
            const slideo_keypoint& fk = fr.kp[v[i].q];
            from[i] = {sk.x, sk.y}; to[i] = {fk.x, fk.y};
        }
        Rated r; r.page = p; r.nvotes = (int)v.size(); r.sim = 0; r.survived = false;
        std::vector<uint8_t> mask(v.size());
        std::fill(r.M, r.M + 9, 0.0);
        if (c.verify_model == 1) r.found = find_homography(from.data(), to.data(), (int)v.size(), c, r.M, mask.data());
        else {
            r.found = estimate_affine_partial(from.data(), to.data(), (int)v.size(), c, r.M, mask.data(), nullptr);
            if (r.found) r.M[8] = 1.0;                               // the 2x3 as a 3x3 (trace only)
        }
        int inl = 0; for (uint8_t m : mask) inl += m;
        r.rating = (double)inl;
        rated.push_back(r);
    }
    std::vector<Rated> all = rated;
    std::stable_sort(rated.begin(), rated.end(), [](const Rated& a, const Rated& b) { return a.rating > b.rating; });  // :329
    if ((int)rated.size() > c.max_rated) rated.resize(c.max_rated);                                                      // :330
    double best_rating = rated.empty() ? 0.0 : rated[0].rating;
    std::vector<Rated> surv;
    for (const Rated& r : rated)
        if (r.rating > c.min_rating && r.rating / best_rating > c.min_rating_ratio) surv.push_back(r);   // :333
    for (Rated& r : surv) {                                    // mo/lib.rs:335-351
        const Page& pg = db.pages[r.page];
        std::vector<uint8_t> proj_small((size_t)pg.sw * pg.sh * 3);
        WarpSampler ws; std::memcpy(ws.M, r.M, sizeof(ws.M)); ws.variant = c.ocv.warp;
        if (c.verify_model == 1) { ws.persp = 1; ws.set_dst_size(pg.w, pg.h); }
        // warp to the slide's size, then to_small_image of that (fused; identical arithmetic)
        int sw, sh; small_size(pg.w, pg.h, c.small_area, sw, sh);
        bool ok = resize_area_generic(pg.w, pg.h, sw, sh, proj_small.data(), [&](int x, int y, uint8_t* p) {
            int sx, sy; ws.src_xy(x, y, sx, sy);
            if ((unsigned)sx < (unsigned)w && (unsigned)sy < (unsigned)h) {
                const uint8_t* s = bgr + (size_t)sy * stride + 3 * sx;
                p[0] = s[0]; p[1] = s[1]; p[2] = s[2];
            } else p[0] = p[1] = p[2] = 0;
        }, c.ocv.area);
        r.sim = ok ? similarity_bgr8(proj_small.data(), pg.small.data(), pg.sw, pg.sh) : 0.f;
        r.survived = true;
    }
    std::vector<Rated> fin = surv;
    // verdict_rule 1 (opt-in departure, include/slideo_amd.h): the survivors keep their rating order (:329) and the similarity
    // only accepts; 0 = the reference: sorted by similarity
    if (c.verdict_rule == 0) std::stable_sort(fin.begin(), fin.end(), [](const Rated& a, const Rated& b) { return a.sim > b.sim; });  // :370
    for (const Rated& r : fin)
        if (r.sim > c.min_similarity) {                        // :381
            out.page_idx = r.page; out.similarity = r.sim; out.inliers = (int)r.rating;
            break;
        }
    if (trace) {
        for (const Rated& a : all) {
            slideo_candidate sc;
            sc.page_idx = a.page; sc.n_votes = a.nvotes; sc.inliers = (int)a.rating;
            sc.survived = 0; sc.similarity = 0;
            std::memcpy(sc.transform, a.M, sizeof(a.M));
            for (const Rated& s : surv) if (s.page == a.page) { sc.survived = 1; sc.similarity = s.sim; }
            trace->push_back(sc);
        }
    }
}

}  // namespace

// ===========================================================================
// C interface (ctypes)
// ===========================================================================
extern "C" {

void so_config_default(slideo_config* c) {
    c->nfeatures = 2000; c->scale_factor = 1.2f; c->nlevels = 8; c->edge_threshold = 62;
    c->patch_size = 62; c->fast_threshold = 20;                    // mo/feature_extractor.rs:14-22
    c->knn_k = 30;                                                 // mo/lib.rs:266
    c->vote_tolerance = 1.05f;                                     // mo/lib.rs:275
    c->max_candidate_pages = 40;                                   // mo/lib.rs:295
    c->ransac_threshold = 3.0; c->ransac_max_iters = 2000; c->ransac_confidence = 0.99;
    c->refine_iters = 10;                                          // mo/image_utils.rs:52
    c->max_rated = 10; c->min_rating = 50.0; c->min_rating_ratio = 0.2;   // mo/lib.rs:330,333
    c->min_similarity = 0.5f;                                      // mo/lib.rs:381
    c->small_area = 300 * 400;                                     // mo/image_utils.rs:11
    c->changed_similarity = 0.98f;                                 // mo/video_capture.rs:98
    c->ratio_test = 0.0f;                                          // extension, off
    c->verify_model = 0;                                           // the reference's estimateAffinePartial2D
    c->matcher = 0; c->lsh_tables = 6; c->lsh_key_bits = 12; c->lsh_multi_probe = 1;   // exact search; mo/flann.rs:16-18
    c->verdict_rule = 0;                                           // mo/lib.rs:370-389: the best re-projection similarity wins
    std::memset(&c->ocv, 0, sizeof(c->ocv));                       // every OpenCV-variant switch at its default (0)
    c->ocv.rng_mul = 4164903690u;                                  // CV_RNG_COEFF
    c->ocv.hdlt = 1;                                               // verify_model 1 only (no reference counterpart): the elimination form, as the library's default
}

int so_config_supported(const slideo_config* c) { return config_supported(*c) ? 1 : 0; }

uint32_t so_rng_next(uint64_t* state) { CvRng r(*state); uint32_t v = r.next(); *state = r.state; return v; }
int so_rng_uniform(uint64_t* state, int a, int b) { CvRng r(*state); int v = r.uniform(a, b); *state = r.state; return v; }

void so_brief_pattern(int patch_size, int32_t* out1024) {
    std::vector<int32_t> p; brief_pattern(patch_size, 512, p);
    std::memcpy(out1024, p.data(), 1024 * sizeof(int32_t));
}
void so_umax(int half_patch, int32_t* out) {
    std::vector<int> u; umax_table(half_patch, u);
    for (int i = 0; i <= half_patch + 1; ++i) out[i] = u[i];
}
void so_level_quotas(const slideo_config* c, int32_t* out) {
    std::vector<int> q; level_quotas(*c, q);
    for (int i = 0; i < c->nlevels; ++i) out[i] = q[i];
}
void so_pyramid_sizes(int w, int h, const slideo_config* c, int32_t* ws, int32_t* hs, float* scales) {
    std::vector<int> a, b; std::vector<float> s; pyramid_sizes(w, h, *c, a, b, s);
    for (int i = 0; i < c->nlevels; ++i) { ws[i] = a[i]; hs[i] = b[i]; scales[i] = s[i]; }
}
void so_gauss_kernel(int n, double sigma, int32_t* out) {
    std::vector<int> k; gauss_kernel_fixed(n, sigma, k);
    for (int i = 0; i < n; ++i) out[i] = k[i];
}
void so_gray_bgr8(const uint8_t* bgr, int w, int h, int stride, uint8_t* out) {
    Img8 g; gray_bgr8(bgr, w, h, stride, g); std::memcpy(out, g.d.data(), g.d.size());
}
void so_resize_linear_exact(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    Img8 s(sw, sh), d; std::memcpy(s.d.data(), src, s.d.size());
    resize_linear_exact(s, dw, dh, d); std::memcpy(dst, d.d.data(), d.d.size());
}
void so_fast_score_map(const uint8_t* img, int w, int h, int thr, uint8_t* out) {
    Img8 s(w, h), sc; std::memcpy(s.d.data(), img, s.d.size());
    fast_score_map(s, thr, sc); std::memcpy(out, sc.d.data(), sc.d.size());
}
// NMS'ed corners as a map (0 or score)
void so_fast_nms_map(const uint8_t* img, int w, int h, int thr, uint8_t* out) {
    Img8 s(w, h), sc; std::memcpy(s.d.data(), img, s.d.size());
    fast_score_map(s, thr, sc);
    std::vector<RawCorner> cs; fast_nms(sc, cs);
    std::memset(out, 0, (size_t)w * h);
    for (const RawCorner& r : cs) out[(size_t)r.y * w + r.x] = (uint8_t)r.score;
}
void so_gaussian_blur7(const uint8_t* img, int w, int h, uint8_t* out) {
    Img8 s(w, h), d; std::memcpy(s.d.data(), img, s.d.size());
    gaussian_blur7(s, d, 3); std::memcpy(out, d.d.data(), d.d.size());    // (GaussianBlur's bit-exact fixed-point path = ocv.blur 3)
}
float so_fast_atan2(float y, float x) { return fast_atan2(y, x); }
// the same primitives with an explicit slideo_ocv_variants value (tests/test_oracle_variants.py, tests/test_opencv_pin.py)
float so_fast_atan2_v(float y, float x, int variant) { return fast_atan2(y, x, variant); }
void so_gray_bgr8_v(const uint8_t* bgr, int w, int h, int stride, uint8_t* out, int variant) {
    Img8 g; gray_bgr8(bgr, w, h, stride, g, variant); std::memcpy(out, g.d.data(), g.d.size());
}
void so_resize_linear_exact_v(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int variant) {
    Img8 s(sw, sh), d; std::memcpy(s.d.data(), src, s.d.size());
    resize_linear_exact(s, dw, dh, d, variant); std::memcpy(dst, d.d.data(), d.d.size());
}
void so_gaussian_blur7_v(const uint8_t* img, int w, int h, uint8_t* out, int variant) {
    Img8 s(w, h), d; std::memcpy(s.d.data(), img, s.d.size());
    gaussian_blur7(s, d, variant); std::memcpy(out, d.d.data(), d.d.size());
}
// integer taps of blur variants 2 / 3, f32 taps of variants 0 / 1
void so_gauss_taps_q8(int variant, int32_t* out7) {
    std::vector<int> k; if (variant == 2) gauss_kernel_q8_rounded(7, 2.0, k); else gauss_kernel_fixed(7, 2.0, k);
    for (int i = 0; i < 7; ++i) out7[i] = k[i];
}
void so_gauss_taps_f32(float* out7) { std::vector<float> k; gauss_kernel_f32(7, 2.0, k); for (int i = 0; i < 7; ++i) out7[i] = k[i]; }
void so_warp_affine_nn_bgr8_v(const uint8_t* src, int sw, int sh, int sstride, const double* M6, uint8_t* dst, int dw, int dh, int variant) {
    warp_affine_nn_bgr8(src, sw, sh, sstride, M6, dst, dw, dh, variant);
}
int so_resize_area_bgr8_v(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int variant) {
    return resize_area_generic(sw, sh, dw, dh, dst, [&](int x, int y, uint8_t* p) {
        const uint8_t* s = src + (size_t)y * sstride + 3 * x; p[0] = s[0]; p[1] = s[1]; p[2] = s[2];
    }, variant) ? 0 : 5;
}
int so_solve4(const double* A16, const double* b4, double* x4, int lm_variant) { return solve4_v(A16, b4, x4, lm_variant) ? 1 : 0; }

int so_pyramid_level(const uint8_t* bgr, int w, int h, int stride, const slideo_config* c, int level,
                     int blurred, uint8_t* out, int64_t cap, int32_t* lw, int32_t* lh) {
    if (level < 0 || level >= c->nlevels) return 1;
    Pyramid p; build_pyramid(bgr, w, h, stride, *c, p, blurred != 0);
    const Img8& im = blurred ? p.blurred[level] : p.level[level];
    *lw = im.w; *lh = im.h;
    if ((int64_t)im.d.size() > cap) return 7;
    std::memcpy(out, im.d.data(), im.d.size());
    return 0;
}

// ORB: returns number of keypoints (writes at most cap)
int so_orb_bgr8(const uint8_t* bgr, int w, int h, int stride, const slideo_config* c,
                slideo_keypoint* kp, uint8_t* desc, int cap) {
    if (!config_supported(*c)) return -1;
    OrbResult r; orb_detect_describe(bgr, w, h, stride, *c, r);
    int n = (int)r.kp.size(), m = std::min(n, cap);
    if (kp) std::memcpy(kp, r.kp.data(), (size_t)m * sizeof(slideo_keypoint));
    if (desc) std::memcpy(desc, r.desc.data(), (size_t)m * 32);
    return n;
}

void so_knn_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int k, int32_t* idx, uint16_t* dist) {
    knn_hamming(q, nq, t, nt, k, idx, dist);
}
// slideo_config.matcher 1 on plain arrays: the k nearest LSH candidates of each query; bits_out (may be null): the tables' bit positions
void so_knn_lsh(const uint8_t* q, int nq, const uint8_t* t, int nt, int k, const slideo_config* c, int32_t* idx, uint16_t* dist, int32_t* bits_out) {
    LshIdx L; L.build(*c, t, nt);
    for (int i = 0; i < nq; ++i) L.knn(q + (size_t)i * 32, t, k, idx + (size_t)i * k, dist + (size_t)i * k);
    if (bits_out) for (size_t i = 0; i < L.bit.size(); ++i) bits_out[i] = L.bit[i];
}
// the cache-blocked / vectorised form the frame path uses; simd: 1 = AVX-512 VPOPCNTDQ was used
int so_knn_hamming_blocked(const uint8_t* q, int nq, const uint8_t* t, int nt, int k, int32_t* idx, uint16_t* dist) {
    std::vector<uint64_t> tb; knn_block_train(t, nt, tb);
    knn_hamming_blocked(q, nq, tb.data(), nt, k, idx, dist);
    return cpu_has_vpopcntdq() ? 1 : 0;
}

// Exact squared-L2 k-NN between 128-dimensional u8 descriptors (SURVEY §8(d) cfg2 / §8(f) N4 — a north-star extension
// with no counterpart in the reference: this brute force IS the parity target of slideo_knn_l2_u8).  k smallest
// (distance, row) pairs per query, ties to the lower row; missing neighbours: idx -1, dist 0xFFFFFFFF.
void so_knn_l2_u8(const uint8_t* q, int nq, const uint8_t* t, int nt, int k, int32_t* idx, uint32_t* dist) {
    std::vector<std::pair<uint32_t, int32_t>> cand;
    for (int i = 0; i < nq; ++i) {
        cand.clear();
        const uint8_t* a = q + (size_t)i * 128;
        for (int j = 0; j < nt; ++j) {
            const uint8_t* b = t + (size_t)j * 128;
            uint32_t d = 0;
            for (int c = 0; c < 128; ++c) { int e = (int)a[c] - (int)b[c]; d += (uint32_t)(e * e); }
            cand.emplace_back(d, j);
        }
        const int kk = std::min(k, nt);
        std::partial_sort(cand.begin(), cand.begin() + kk, cand.end());
        for (int r = 0; r < k; ++r) {
            idx[(size_t)i * k + r] = r < kk ? cand[r].second : -1;
            dist[(size_t)i * k + r] = r < kk ? cand[r].first : 0xFFFFFFFFu;
        }
    }
}

int so_estimate_affine_partial(const float* from_xy, const float* to_xy, int n, const slideo_config* c,
                               double* M6, uint8_t* mask, int32_t* iters_run) {
    int it = 0;
    bool f = estimate_affine_partial((const P2f*)from_xy, (const P2f*)to_xy, n, *c, M6, mask, &it);
    if (iters_run) *iters_run = it;
    return f ? 1 : 0;
}

// verify_model 1 primitives (tests/test_oracle_homography.py)
int so_find_homography(const float* from_xy, const float* to_xy, int n, const slideo_config* c, double* H9, uint8_t* mask,
                       int32_t* stats4 /* iterations, subset attempts, Jacobi rotations, RNG draws; may be null */) {
    HomographyStats st;
    bool f = find_homography((const P2f*)from_xy, (const P2f*)to_xy, n, *c, H9, mask, &st);
    if (stats4) { stats4[0] = st.iters; stats4[1] = st.attempts; stats4[2] = st.rotations; stats4[3] = st.draws; }
    return f ? 1 : 0;
}
int so_homography_dlt(const float* from_xy, const float* to_xy, int n, double* H9, int variant) {
    if (variant == 1 && n == 4) return homography_4pt_direct((const P2f*)from_xy, (const P2f*)to_xy, H9);
    if (variant == 2 && n == 4) return homography_4pt_closed((const P2f*)from_xy, (const P2f*)to_xy, H9);
    return homography_dlt((const P2f*)from_xy, (const P2f*)to_xy, n, H9);
}
int so_homography_check_subset(const float* from_xy, const float* to_xy, int n) {
    return homography_check_subset((const P2f*)from_xy, (const P2f*)to_xy, n) ? 1 : 0;
}
int so_jacobi_eig(const double* A, int n, double* W, double* V) {
    if (n < 1 || n > 16) return -1;
    std::vector<double> a(A, A + n * n);
    return jacobi_eig_n(a.data(), n, W, V);
}
void so_warp_perspective_nn_bgr8(const uint8_t* src, int sw, int sh, int sstride, const double* H9, uint8_t* dst, int dw, int dh) {
    warp_perspective_nn_bgr8(src, sw, sh, sstride, H9, dst, dw, dh);
}

void so_warp_affine_nn_bgr8(const uint8_t* src, int sw, int sh, int sstride, const double* M6,
                            uint8_t* dst, int dw, int dh) {
    warp_affine_nn_bgr8(src, sw, sh, sstride, M6, dst, dw, dh);
}

int so_resize_area_bgr8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh) {
    return resize_area_generic(sw, sh, dw, dh, dst, [&](int x, int y, uint8_t* p) {
        const uint8_t* s = src + (size_t)y * sstride + 3 * x; p[0] = s[0]; p[1] = s[1]; p[2] = s[2];
    }) ? 0 : 5;
}

void so_small_size(int w, int h, int small_area, int32_t* sw, int32_t* sh) {
    int a, b; small_size(w, h, small_area, a, b); *sw = a; *sh = b;
}
int so_small_image_bgr8(const uint8_t* bgr, int w, int h, int stride, int small_area, uint8_t* out, int64_t cap,
                        int32_t* sw, int32_t* sh) {
    std::vector<uint8_t> o; int a, b;
    if (!small_image(bgr, w, h, stride, small_area, o, a, b)) return 5;
    *sw = a; *sh = b;
    if ((int64_t)o.size() > cap) return 7;
    std::memcpy(out, o.data(), o.size());
    return 0;
}
float so_similarity_bgr8(const uint8_t* a, const uint8_t* b, int w, int h) { return similarity_bgr8(a, b, w, h); }

// mo/video_capture.rs:86-98 MarkSimilarIter over n frames
int so_changed_mask_bgr8(const uint8_t* frames, int n, int w, int h, int stride, int64_t frame_stride,
                         const slideo_config* c, const uint8_t* prev_small, uint8_t* last_small_out,
                         uint8_t* changed, float* sims) {
    std::vector<uint8_t> last, cur; int sw = 0, sh = 0;
    bool have = false;
    if (prev_small) {
        small_size(w, h, c->small_area, sw, sh);
        last.assign(prev_small, prev_small + (size_t)sw * sh * 3); have = true;
    }
    for (int i = 0; i < n; ++i) {
        if (!small_image(frames + (size_t)i * frame_stride, w, h, stride, c->small_area, cur, sw, sh, c->ocv.area)) return 5;
        float s = have ? similarity_bgr8(last.data(), cur.data(), sw, sh) : 0.0f;
        changed[i] = s < c->changed_similarity;
        if (sims) sims[i] = s;
        last.swap(cur); have = true;
    }
    if (last_small_out && have) std::memcpy(last_small_out, last.data(), last.size());
    return 0;
}

so_pagedb* so_pagedb_create(const slideo_config* c) {
    if (!config_supported(*c)) return nullptr;
    so_pagedb* db = new so_pagedb(); db->cfg = *c; return db;
}
void so_pagedb_destroy(so_pagedb* db) { delete db; }
// the product's slideo_matcher_use_sift: before the first page
int so_pagedb_use_sift(so_pagedb* db, const so_sift_config* sc, float ratio) {
    if (!db->pages.empty() || db->finalized) return 4;
    if (sc->n_octave_layers != 3 || !(ratio >= 0.f) || !(ratio <= 1.f) || db->cfg.matcher != 0) return 1;
    db->sift = true; db->sc = *sc; db->ratio = ratio;
    return 0;
}

int so_pagedb_add_page(so_pagedb* db, const uint8_t* bgr, int w, int h, int stride) {   // mo/lib.rs:92-131
    if (db->finalized) return 4;
    Page p; p.w = w; p.h = h;
    if (db->sift) { SiftResult sr; sift_detect_describe(bgr, w, h, stride, db->sc, db->cfg.ocv, sr); p.orb.kp = sr.kp; p.sdesc = sr.desc; }
    else
    orb_detect_describe(bgr, w, h, stride, db->cfg, p.orb);
    if (!small_image(bgr, w, h, stride, db->cfg.small_area, p.small, p.sw, p.sh, db->cfg.ocv.area)) return 5;
    db->pages.push_back(std::move(p));
    return 0;
}
// n equally sized pages, `threads` workers (mirrors rayon par_iter over pages, mo/lib.rs:45-47)
int so_pagedb_add_pages(so_pagedb* db, const uint8_t* pages, int n, int w, int h, int stride, int64_t page_stride, int threads) {
    if (db->finalized) return 4;
    size_t base = db->pages.size();
    db->pages.resize(base + n);
    threads = std::max(1, std::min(threads, n));
    std::vector<int> rc(threads, 0);
    auto work = [&](int t) {
        for (int i = t; i < n; i += threads) {
            Page& p = db->pages[base + i];
            p.w = w; p.h = h;
            const uint8_t* img = pages + (size_t)i * page_stride;
            if (db->sift) { SiftResult sr; sift_detect_describe(img, w, h, stride, db->sc, db->cfg.ocv, sr); p.orb.kp = sr.kp; p.sdesc = sr.desc; }
            else
            orb_detect_describe(img, w, h, stride, db->cfg, p.orb);
            if (!small_image(img, w, h, stride, db->cfg.small_area, p.small, p.sw, p.sh, db->cfg.ocv.area)) rc[t] = 5;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
    for (int r : rc) if (r) return r;
    return 0;
}
int so_pagedb_finalize(so_pagedb* db) {   // mo/flann.rs:65-71 (exact index = the concatenation)
    db->train.clear(); db->train128.clear(); db->train_page.clear(); db->page_ofs.assign(1, 0);
    for (size_t p = 0; p < db->pages.size(); ++p) {
        const OrbResult& o = db->pages[p].orb;
        db->train.insert(db->train.end(), o.desc.begin(), o.desc.end());
        db->train128.insert(db->train128.end(), db->pages[p].sdesc.begin(), db->pages[p].sdesc.end());
        db->train_page.insert(db->train_page.end(), o.kp.size(), (int32_t)p);
        db->page_ofs.push_back((int32_t)db->train_page.size());
    }
    if (!db->sift) {
        knn_block_train(db->train.data(), (int)db->train_page.size(), db->train_blocked);
        if (db->cfg.matcher == 1) db->lsh.build(db->cfg, db->train.data(), (int)db->train_page.size());
    }
    db->finalized = true;
    return db->train_page.empty() ? 6 : 0;
}
int64_t so_pagedb_descriptor_count(const so_pagedb* db) { return (int64_t)db->train_page.size(); }
int so_pagedb_page_count(const so_pagedb* db) { return (int)db->pages.size(); }
int so_pagedb_get_page_features(const so_pagedb* db, int page, slideo_keypoint* kp, uint8_t* desc, int cap) {
    const OrbResult& o = db->pages[page].orb;
    int n = (int)o.kp.size(), m = std::min(n, cap);
    if (kp) std::memcpy(kp, o.kp.data(), (size_t)m * sizeof(slideo_keypoint));
    if (desc && db->sift) std::memcpy(desc, db->pages[page].sdesc.data(), (size_t)m * 128);
    else if (desc) std::memcpy(desc, o.desc.data(), (size_t)m * 32);
    return n;
}
int so_pagedb_get_train(const so_pagedb* db, uint8_t* train, int64_t cap_rows) {
    int64_t m = (int64_t)db->train_page.size();
    if (m > cap_rows) return 7;
    std::memcpy(train, db->train.data(), (size_t)m * 32);
    return 0;
}

// match n frames (threads > 1: one frame per worker, mirrors rayon spawn_fifo mo/lib.rs:213)
int so_match_frames(const so_pagedb* db, const uint8_t* frames, int n, int w, int h, int stride,
                    int64_t frame_stride, slideo_verdict* out, int threads) {
    if (!db->finalized || db->train_page.empty()) return 6;
    threads = std::max(1, std::min(threads, n));
    if (threads == 1) {
        for (int i = 0; i < n; ++i) match_frame(*db, frames + (size_t)i * frame_stride, w, h, stride, out[i], nullptr);
        return 0;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([=]() {
            for (int i = t; i < n; i += threads)
                match_frame(*db, frames + (size_t)i * frame_stride, w, h, stride, out[i], nullptr);
        });
    for (auto& x : th) x.join();
    return 0;
}

// single frame with the decision trace
int so_match_frame_trace(const so_pagedb* db, const uint8_t* bgr, int w, int h, int stride,
                         slideo_verdict* out, slideo_candidate* cands, int cap, int32_t* n_cands) {
    if (!db->finalized || db->train_page.empty()) return 6;
    std::vector<slideo_candidate> tr;
    match_frame(*db, bgr, w, h, stride, *out, &tr);
    *n_cands = (int32_t)tr.size();
    int m = std::min((int)tr.size(), cap);
    std::memcpy(cands, tr.data(), (size_t)m * sizeof(slideo_candidate));
    return 0;
}

// mo/lib.rs:229-244: sort by time (stable), drop consecutive equal `image`.
// in: n records (time_ms, frame_idx, page) — out: indices kept, returns count.
int so_timeline_dedup(const int64_t* time_ms, const int32_t* page, int n, int32_t* keep_idx) {
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return time_ms[a] < time_ms[b]; });
    int m = 0; bool have = false; int last = 0;
    for (int i : order) {
        if (have && last == page[i]) continue;
        last = page[i]; have = true;
        keep_idx[m++] = i;
    }
    return m;
}

}  // extern "C"
