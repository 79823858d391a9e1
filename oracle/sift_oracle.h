/*
 * sift_oracle.h — TEST INFRASTRUCTURE (part of liboracle.so; included by slideo_oracle.cpp after its helpers).
 *
 * CPU restatement of cv::SIFT::detectAndCompute of OpenCV 4.5.2 (features2d/src/sift.dispatch.cpp, sift.simd.hpp), RECALLED:
 * BASELINE configs[2] / north_star name "SIFT keypoint detect+describe", which the reference never calls (its only extractor is
 * ORB, crates/matching-opencv/src/feature_extractor.rs:3-4,13 — SURVEY F6, section 8(f) N4).  There is therefore no reference
 * call site, test or golden vector to anchor on: the parity target of the HIP extractor (csrc/sift.hip.h) is THIS restatement.
 *
 *   createInitialImage    BGR -> gray (8 bit, the ORB path's cvtColor) -> f32, 2x bilinear upsample (firstOctave = -1),
 *                         GaussianBlur(sig_diff = sqrt(sigma^2 - 4 * 0.5^2))
 *   buildGaussianPyramid  nOctaves = cvRound(log2(min side of the doubled image) - 2) + 1, nOctaveLayers + 3 layers per octave,
 *                         layer i = GaussianBlur(layer i - 1, sig[i]); octave o + 1 starts from layer nOctaveLayers of octave o,
 *                         every second pixel (resize INTER_NEAREST)
 *   buildDoGPyramid       differences of adjacent layers
 *   findScaleSpaceExtrema |v| > floor(0.5 * contrastThreshold / nOctaveLayers * 255), >= / <= all 26 neighbours; adjustLocalExtrema
 *                         (<= 5 Newton steps on the 3-D quadratic, Matx33f::solve = Cramer's rule in f32, contrast and edge tests);
 *                         orientation histogram (36 bins, radius cvRound(4.5 scl), weights exp(-(i^2+j^2) / (2 (1.5 scl)^2)),
 *                         fastAtan2, smoothing 1 4 6 4 1, peaks >= 0.8 max, parabolic interpolation)
 *   removeDuplicated, retainBest(nfeatures) by response (ties kept), keypoints back to the input scale (x 0.5)
 *   calcSIFTDescriptor    4 x 4 x 8 histogram, trilinear, Gaussian window, clamp at 0.2 |v|, x 512 / |v|, saturate to u8
 *
 * Two deliberate, documented departures (both are float-summation-order matters; neither is observable above the descriptor's
 * 8-bit quantisation except at rounding boundaries):
 *   - the orientation and descriptor histograms are accumulated in FIXED POINT (contribution -> llrint(v * 2^20), i64 sums):
 *     order independent, so that a GPU that scatters with integer atomics reproduces them bit for bit; OpenCV adds f32 in
 *     sample order.  (exp, cos, sin: evaluated in f64 and rounded to f32 on both sides; OpenCV's exp32f is its own table-based
 *     routine.)
 *   - canonical keypoint order: (octave, layer, row, column of the refined extremum, histogram bin of the peak) ascending.
 * GaussianBlur on f32 = sepFilter2D with getGaussianKernel(cvRound(8 sigma + 1) | 1, sigma) in f32, BORDER_REFLECT_101; its
 * accumulation order and FMA contraction follow slideo_ocv_variants.blur (0: contracted, 1: not — the same switch as ORB's).
 */
#pragma once

struct so_sift_config {              /* mirror of slideo_sift_config (include/slideo_amd.h) */
    int32_t nfeatures;               /* 0 = keep all */
    int32_t n_octave_layers;         /* 3 */
    double contrast_threshold;       /* 0.04 */
    double edge_threshold;           /* 10 */
    double sigma;                    /* 1.6 */
};

namespace {

struct ImgF {
    int w = 0, h = 0;
    std::vector<float> d;
    ImgF() {}
    ImgF(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
    float at(int y, int x) const { return d[(size_t)y * w + x]; }
    float& at(int y, int x) { return d[(size_t)y * w + x]; }
};

constexpr int SIFT_IMG_BORDER = 5, SIFT_MAX_INTERP_STEPS = 5, SIFT_ORI_HIST_BINS = 36;
constexpr float SIFT_ORI_SIG_FCTR = 1.5f, SIFT_ORI_RADIUS = 4.5f, SIFT_ORI_PEAK_RATIO = 0.8f, SIFT_DESCR_SCL_FCTR = 3.f,
                SIFT_DESCR_MAG_THR = 0.2f, SIFT_INT_DESCR_FCTR = 512.f, SIFT_INIT_SIGMA = 0.5f;
constexpr double SIFT_FIX = 1048576.0;       /* 2^20: resolution of the fixed-point histogram sums */

static inline float exp_f32(float x) { return (float)std::exp((double)x); }

static void sift_gauss_kernel(double sigma, std::vector<float>& k) {
    int n = cv_round(sigma * 4 * 2 + 1) | 1;
    k.resize(n);
    std::vector<double> kd(n);
    double sum = 0, s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; ++i) { double x = i - (n - 1) * 0.5; kd[i] = std::exp(s2 * x * x); sum += kd[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) k[i] = (float)(kd[i] * sum);
}

// GaussianBlur(src, dst, Size(), sigma) on CV_32F: RowFilter (taps in order) then SymmColumnFilter (centre, then symmetric pairs)
static void sift_blur(const ImgF& src, ImgF& dst, double sigma, bool fma) {
    std::vector<float> k; sift_gauss_kernel(sigma, k);
    const int n = (int)k.size(), r = n / 2, w = src.w, h = src.h;
    ImgF tmp(w, h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = k[0] * src.at(y, reflect101(x - r, w));
            for (int j = 1; j < n; ++j) s = mad_f32(k[j], src.at(y, reflect101(x - r + j, w)), s, fma);
            tmp.at(y, x) = s;
        }
    dst = ImgF(w, h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = k[r] * tmp.at(y, x);
            for (int j = 1; j <= r; ++j) s = mad_f32(k[r + j], tmp.at(reflect101(y + j, h), x) + tmp.at(reflect101(y - j, h), x), s, fma);
            dst.at(y, x) = s;
        }
}

// resize(gray_f32, 2x, INTER_LINEAR): horizontal pass then vertical pass, half-pixel centres, clamped at the border
static void sift_upsample2(const ImgF& src, ImgF& dst) {
    const int w = src.w, h = src.h, W = 2 * w, H = 2 * h;
    auto coef = [](int d, int n, int& i0, int& i1, float& a0, float& a1) {
        float f = (float)((d + 0.5) * 0.5 - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (s < 0) { s = 0; f = 0; }
        if (s >= n - 1) { s = n - 1; f = 0; }
        i0 = s; i1 = std::min(s + 1, n - 1); a0 = 1.f - f; a1 = f;
    };
    ImgF tmp(W, h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < W; ++x) { int i0, i1; float a0, a1; coef(x, w, i0, i1, a0, a1); tmp.at(y, x) = src.at(y, i0) * a0 + src.at(y, i1) * a1; }
    dst = ImgF(W, H);
    for (int y = 0; y < H; ++y) {
        int i0, i1; float b0, b1; coef(y, h, i0, i1, b0, b1);
        for (int x = 0; x < W; ++x) dst.at(y, x) = tmp.at(i0, x) * b0 + tmp.at(i1, x) * b1;
    }
}

struct SiftPyr {
    int n_oct = 0, nl = 3;
    std::vector<ImgF> g, dog;          /* g[o * (nl + 3) + i], dog[o * (nl + 2) + i] */
};

static int sift_num_octaves(int bw, int bh) { return cv_round(std::log((double)std::min(bw, bh)) / std::log(2.) - 2) + 1; }

static void sift_build(const uint8_t* bgr, int w, int h, int stride, const so_sift_config& sc, const slideo_ocv_variants& ocv, SiftPyr& P) {
    Img8 gray; gray_bgr8(bgr, w, h, stride, gray, ocv.gray);
    ImgF gf(w, h);
    for (size_t i = 0; i < gf.d.size(); ++i) gf.d[i] = (float)gray.d[i];
    ImgF dbl; sift_upsample2(gf, dbl);
    const bool fma = ocv.blur != 1;
    const float sigma = (float)sc.sigma;
    const float sig_diff = std::sqrt(std::max(sigma * sigma - SIFT_INIT_SIGMA * SIFT_INIT_SIGMA * 4, 0.01f));
    ImgF base; sift_blur(dbl, base, sig_diff, fma);
    const int nl = sc.n_octave_layers;
    P.nl = nl; P.n_oct = std::max(sift_num_octaves(base.w, base.h), 0);
    std::vector<double> sig(nl + 3);
    sig[0] = sc.sigma;
    const double k = std::pow(2., 1. / nl);
    for (int i = 1; i < nl + 3; ++i) { double sp = std::pow(k, (double)(i - 1)) * sc.sigma, st = sp * k; sig[i] = std::sqrt(st * st - sp * sp); }
    P.g.assign((size_t)P.n_oct * (nl + 3), ImgF());
    for (int o = 0; o < P.n_oct; ++o)
        for (int i = 0; i < nl + 3; ++i) {
            ImgF& dst = P.g[(size_t)o * (nl + 3) + i];
            if (o == 0 && i == 0) dst = base;
            else if (i == 0) {
                const ImgF& src = P.g[(size_t)(o - 1) * (nl + 3) + nl];
                dst = ImgF(src.w / 2, src.h / 2);
                for (int y = 0; y < dst.h; ++y) for (int x = 0; x < dst.w; ++x) dst.at(y, x) = src.at(2 * y, 2 * x);
            } else sift_blur(P.g[(size_t)o * (nl + 3) + i - 1], dst, sig[i], fma);
        }
    P.dog.assign((size_t)P.n_oct * (nl + 2), ImgF());
    for (int o = 0; o < P.n_oct; ++o)
        for (int i = 0; i < nl + 2; ++i) {
            const ImgF& a = P.g[(size_t)o * (nl + 3) + i];
            const ImgF& b = P.g[(size_t)o * (nl + 3) + i + 1];
            ImgF& d = P.dog[(size_t)o * (nl + 2) + i];
            d = ImgF(a.w, a.h);
            for (size_t t = 0; t < d.d.size(); ++t) d.d[t] = b.d[t] - a.d[t];
        }
}

struct SiftKp { slideo_keypoint kp; uint64_t key; };      /* key: (octave, layer, r, c, bin) — the canonical order */

// Matx33f::solve(b, DECOMP_LU) of matx.hpp (Matx_FastSolveOp<float, 3, 3, 1>): Cramer's rule in f32; singular -> x = 0
static void solve3_f32(const float a[3][3], const float b[3], float x[3]) {
    float d = a[0][0] * (a[1][1] * a[2][2] - a[2][1] * a[1][2]) - a[0][1] * (a[1][0] * a[2][2] - a[2][0] * a[1][2]) +
              a[0][2] * (a[1][0] * a[2][1] - a[2][0] * a[1][1]);
    x[0] = x[1] = x[2] = 0;
    if (d == 0) return;
    d = 1 / d;
    x[0] = d * (b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2]) + a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
    x[1] = d * (a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) - b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) + a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
    x[2] = d * (a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) - a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0]) + b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
}

// adjustLocalExtrema; returns false when the candidate is rejected.  layer / r / c are updated.
static bool sift_adjust(const SiftPyr& P, const so_sift_config& sc, int octv, int& layer, int& r, int& c, slideo_keypoint& kpt) {
    const int nl = P.nl;
    const float img_scale = 1.f / 255.f, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale, cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0;
    for (; i < SIFT_MAX_INTERP_STEPS; ++i) {
        const ImgF& img = P.dog[(size_t)octv * (nl + 2) + layer];
        const ImgF& prev = P.dog[(size_t)octv * (nl + 2) + layer - 1];
        const ImgF& next = P.dog[(size_t)octv * (nl + 2) + layer + 1];
        const float dD[3] = {(img.at(r, c + 1) - img.at(r, c - 1)) * deriv_scale, (img.at(r + 1, c) - img.at(r - 1, c)) * deriv_scale,
                             (next.at(r, c) - prev.at(r, c)) * deriv_scale};
        const float v2 = img.at(r, c) * 2;
        const float dxx = (img.at(r, c + 1) + img.at(r, c - 1) - v2) * second_deriv_scale;
        const float dyy = (img.at(r + 1, c) + img.at(r - 1, c) - v2) * second_deriv_scale;
        const float dss = (next.at(r, c) + prev.at(r, c) - v2) * second_deriv_scale;
        const float dxy = (img.at(r + 1, c + 1) - img.at(r + 1, c - 1) - img.at(r - 1, c + 1) + img.at(r - 1, c - 1)) * cross_deriv_scale;
        const float dxs = (next.at(r, c + 1) - next.at(r, c - 1) - prev.at(r, c + 1) + prev.at(r, c - 1)) * cross_deriv_scale;
        const float dys = (next.at(r + 1, c) - next.at(r - 1, c) - prev.at(r + 1, c) + prev.at(r - 1, c)) * cross_deriv_scale;
        const float H[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        float X[3];
        solve3_f32(H, dD, X);
        xi = -X[2]; xr = -X[1]; xc = -X[0];
        if (std::fabs(xi) < 0.5f && std::fabs(xr) < 0.5f && std::fabs(xc) < 0.5f) break;
        if (std::fabs(xi) > (float)(INT32_MAX / 3) || std::fabs(xr) > (float)(INT32_MAX / 3) || std::fabs(xc) > (float)(INT32_MAX / 3)) return false;
        c += cv_round(xc); r += cv_round(xr); layer += cv_round(xi);
        if (layer < 1 || layer > nl || c < SIFT_IMG_BORDER || c >= img.w - SIFT_IMG_BORDER || r < SIFT_IMG_BORDER || r >= img.h - SIFT_IMG_BORDER) return false;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return false;
    {
        const ImgF& img = P.dog[(size_t)octv * (nl + 2) + layer];
        const ImgF& prev = P.dog[(size_t)octv * (nl + 2) + layer - 1];
        const ImgF& next = P.dog[(size_t)octv * (nl + 2) + layer + 1];
        const float dD[3] = {(img.at(r, c + 1) - img.at(r, c - 1)) * deriv_scale, (img.at(r + 1, c) - img.at(r - 1, c)) * deriv_scale,
                             (next.at(r, c) - prev.at(r, c)) * deriv_scale};
        const float t = dD[0] * xc + dD[1] * xr + dD[2] * xi;
        const float contr = img.at(r, c) * img_scale + t * 0.5f;
        if (std::fabs(contr) * nl < (float)sc.contrast_threshold) return false;
        const float v2 = img.at(r, c) * 2.f;
        const float dxx = (img.at(r, c + 1) + img.at(r, c - 1) - v2) * second_deriv_scale;
        const float dyy = (img.at(r + 1, c) + img.at(r - 1, c) - v2) * second_deriv_scale;
        const float dxy = (img.at(r + 1, c + 1) - img.at(r + 1, c - 1) - img.at(r - 1, c + 1) + img.at(r - 1, c - 1)) * cross_deriv_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        const float et = (float)sc.edge_threshold;
        if (det <= 0 || tr * tr * et >= (et + 1) * (et + 1) * det) return false;
        kpt.x = (c + xc) * (float)(1 << octv);
        kpt.y = (r + xr) * (float)(1 << octv);
        kpt.octave = octv + (layer << 8) + (cv_round((xi + 0.5) * 255) << 16);
        kpt.size = (float)sc.sigma * (float)std::pow(2.0, (double)((layer + xi) / nl)) * (float)(1 << octv) * 2;      /* (powf evaluated in f64, like exp / cos / sin) */
        kpt.response = std::fabs(contr);
    }
    return true;
}

// calcOrientationHist (fixed-point accumulation, see the header); returns the maximum of the smoothed histogram
static float sift_ori_hist(const ImgF& img, int px, int py, int radius, float sigma, float* hist, int n, int atan_v) {
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    std::vector<int64_t> acc(n, 0);
    for (int i = -radius; i <= radius; ++i) {
        const int y = py + i;
        if (y <= 0 || y >= img.h - 1) continue;
        for (int j = -radius; j <= radius; ++j) {
            const int x = px + j;
            if (x <= 0 || x >= img.w - 1) continue;
            const float dx = img.at(y, x + 1) - img.at(y, x - 1), dy = img.at(y - 1, x) - img.at(y + 1, x);
            const float wgt = exp_f32((float)(i * i + j * j) * expf_scale);
            const float ori = fast_atan2(dy, dx, atan_v), mag = std::sqrt(dx * dx + dy * dy);
            int bin = cv_round((n / 360.f) * ori);
            if (bin >= n) bin -= n;
            if (bin < 0) bin += n;
            acc[bin] += (int64_t)std::llrint((double)(wgt * mag) * SIFT_FIX);
        }
    }
    std::vector<float> th(n + 4);
    for (int i = 0; i < n; ++i) th[i + 2] = (float)((double)acc[i] / SIFT_FIX);
    th[1] = th[n + 1]; th[0] = th[n]; th[n + 2] = th[2]; th[n + 3] = th[3];
    float mx = 0;
    for (int i = 0; i < n; ++i) {
        hist[i] = (th[i] + th[i + 4]) * (1.f / 16.f) + (th[i + 1] + th[i + 3]) * (4.f / 16.f) + th[i + 2] * (6.f / 16.f);
        mx = i == 0 ? hist[0] : std::max(mx, hist[i]);
    }
    return mx;
}

// calcSIFTDescriptor (fixed-point accumulation): 128 values 0..255
static void sift_descriptor(const ImgF& img, float ptx, float pty, float ori, float scl, uint8_t* dst, int atan_v) {
    const int d = 4, n = 8;
    const int px = cv_round(ptx), py = cv_round(pty);
    float cos_t = (float)std::cos((double)(ori * (float)(3.14159265358979323846 / 180)));
    float sin_t = (float)std::sin((double)(ori * (float)(3.14159265358979323846 / 180)));
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = SIFT_DESCR_SCL_FCTR * scl;
    int radius = cv_round(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    radius = std::min(radius, (int)std::sqrt((double)img.w * img.w + (double)img.h * img.h));
    cos_t /= hist_width; sin_t /= hist_width;
    int64_t hist[(4 + 2) * (4 + 2) * (8 + 2)];
    std::fill(hist, hist + 360, (int64_t)0);
    for (int i = -radius; i <= radius; ++i)
        for (int j = -radius; j <= radius; ++j) {
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < img.h - 1 && c > 0 && c < img.w - 1)) continue;
            const float dx = img.at(r, c + 1) - img.at(r, c - 1), dy = img.at(r - 1, c) - img.at(r + 1, c);
            const float w = exp_f32((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            const float o = fast_atan2(dy, dx, atan_v);
            float obin = (o - ori) * bins_per_rad;
            const float mag = std::sqrt(dx * dx + dy * dy) * w;
            const int r0 = cv_floor(rbin), c0 = cv_floor(cbin);
            int o0 = cv_floor(obin);
            rbin -= r0; cbin -= c0; obin -= o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111, v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
            const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011, v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
            const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
            auto add = [&](int k, float v) { hist[k] += (int64_t)std::llrint((double)v * SIFT_FIX); };
            add(idx, v_rco000); add(idx + 1, v_rco001); add(idx + (n + 2), v_rco010); add(idx + (n + 3), v_rco011);
            add(idx + (d + 2) * (n + 2), v_rco100); add(idx + (d + 2) * (n + 2) + 1, v_rco101);
            add(idx + (d + 3) * (n + 2), v_rco110); add(idx + (d + 3) * (n + 2) + 1, v_rco111);
        }
    float v[128];
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            const int64_t h0 = hist[idx] + hist[idx + n], h1 = hist[idx + 1] + hist[idx + n + 1];       /* circular orientation bins */
            for (int k = 0; k < n; ++k) v[(i * d + j) * n + k] = (float)((double)(k == 0 ? h0 : k == 1 ? h1 : hist[idx + k]) / SIFT_FIX);
        }
    float nrm2 = 0;
    for (int k = 0; k < 128; ++k) nrm2 += v[k] * v[k];
    const float thr = std::sqrt(nrm2) * SIFT_DESCR_MAG_THR;
    nrm2 = 0;
    for (int k = 0; k < 128; ++k) { const float val = std::min(v[k], thr); v[k] = val; nrm2 += val * val; }
    nrm2 = SIFT_INT_DESCR_FCTR / std::max(std::sqrt(nrm2), FLT_EPSILON);
    for (int k = 0; k < 128; ++k) {
        const int q = (int)std::lrintf(v[k] * nrm2);
        dst[k] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
}

struct SiftResult { std::vector<slideo_keypoint> kp; std::vector<uint8_t> desc; int n_octaves = 0, n_extrema = 0, n_refined = 0; };

static void sift_detect_describe(const uint8_t* bgr, int w, int h, int stride, const so_sift_config& sc, const slideo_ocv_variants& ocv, SiftResult& out) {
    SiftPyr P;
    sift_build(bgr, w, h, stride, sc, ocv, P);
    const int nl = P.nl, n = SIFT_ORI_HIST_BINS;
    out.n_octaves = P.n_oct;
    const int threshold = cv_floor(0.5 * sc.contrast_threshold / nl * 255);
    std::vector<SiftKp> kps;
    for (int o = 0; o < P.n_oct; ++o)
        for (int i = 1; i <= nl; ++i) {
            const ImgF& img = P.dog[(size_t)o * (nl + 2) + i];
            const ImgF& prev = P.dog[(size_t)o * (nl + 2) + i - 1];
            const ImgF& next = P.dog[(size_t)o * (nl + 2) + i + 1];
            for (int r = SIFT_IMG_BORDER; r < img.h - SIFT_IMG_BORDER; ++r)
                for (int c = SIFT_IMG_BORDER; c < img.w - SIFT_IMG_BORDER; ++c) {
                    const float val = img.at(r, c);
                    if (!(std::fabs(val) > threshold)) continue;
                    bool ext = true;
                    if (val > 0) {
                        for (int dy = -1; dy <= 1 && ext; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                            if ((dy || dx) && !(val >= img.at(r + dy, c + dx))) { ext = false; break; }
                            if (!(val >= prev.at(r + dy, c + dx)) || !(val >= next.at(r + dy, c + dx))) { ext = false; break; }
                        }
                    } else {
                        for (int dy = -1; dy <= 1 && ext; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                            if ((dy || dx) && !(val <= img.at(r + dy, c + dx))) { ext = false; break; }
                            if (!(val <= prev.at(r + dy, c + dx)) || !(val <= next.at(r + dy, c + dx))) { ext = false; break; }
                        }
                    }
                    if (!ext) continue;
                    out.n_extrema++;
                    int r1 = r, c1 = c, layer = i;
                    slideo_keypoint kpt{};
                    if (!sift_adjust(P, sc, o, layer, r1, c1, kpt)) continue;
                    out.n_refined++;
                    const float scl_octv = kpt.size * 0.5f / (float)(1 << o);
                    float hist[SIFT_ORI_HIST_BINS];
                    const float omax = sift_ori_hist(P.g[(size_t)o * (nl + 3) + layer], c1, r1, cv_round(SIFT_ORI_RADIUS * scl_octv),
                                                     SIFT_ORI_SIG_FCTR * scl_octv, hist, n, ocv.atan);
                    const float mag_thr = omax * SIFT_ORI_PEAK_RATIO;
                    for (int j = 0; j < n; ++j) {
                        const int l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
                        if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
                            float bin = j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
                            bin = bin < 0 ? n + bin : bin >= n ? bin - n : bin;
                            slideo_keypoint k2 = kpt;
                            k2.angle = 360.f - (float)((360.f / n) * bin);
                            if (std::fabs(k2.angle - 360.f) < FLT_EPSILON) k2.angle = 0.f;
                            const uint64_t key = ((uint64_t)o << 34) | ((uint64_t)layer << 32) | ((uint64_t)r1 << 19) | ((uint64_t)c1 << 6) | (uint64_t)j;
                            kps.push_back({k2, key});
                        }
                    }
                }
        }
    // canonical order; equal keys are exact duplicates (two extrema refined to the same place): removeDuplicated
    std::sort(kps.begin(), kps.end(), [](const SiftKp& a, const SiftKp& b) { return a.key < b.key; });
    kps.erase(std::unique(kps.begin(), kps.end(), [](const SiftKp& a, const SiftKp& b) { return a.key == b.key; }), kps.end());
    // retainBest(nfeatures): every keypoint whose response reaches the n-th largest stays (ties kept)
    if (sc.nfeatures > 0 && (int)kps.size() > sc.nfeatures) {
        std::vector<float> resp;
        for (const SiftKp& k : kps) resp.push_back(k.kp.response);
        std::nth_element(resp.begin(), resp.begin() + (sc.nfeatures - 1), resp.end(), std::greater<float>());
        const float thr = resp[sc.nfeatures - 1];
        kps.erase(std::remove_if(kps.begin(), kps.end(), [&](const SiftKp& k) { return !(k.kp.response >= thr); }), kps.end());
    }
    out.kp.clear(); out.desc.assign(kps.size() * 128, 0);
    for (size_t t = 0; t < kps.size(); ++t) {
        slideo_keypoint k = kps[t].kp;
        // descriptor on the pyramid (before the keypoint is scaled back to the input image: firstOctave = -1)
        const int octave = k.octave & 255, layer = (k.octave >> 8) & 255;
        const float scale = 1.f / (float)(1 << octave);
        const float size = k.size * scale;
        float angle = 360.f - k.angle;
        if (std::fabs(angle - 360.f) < FLT_EPSILON) angle = 0.f;
        sift_descriptor(P.g[(size_t)octave * (nl + 3) + layer], k.x * scale, k.y * scale, angle, size * 0.5f, out.desc.data() + t * 128, ocv.atan);
        k.octave = (k.octave & ~255) | ((k.octave - 1) & 255);
        k.x *= 0.5f; k.y *= 0.5f; k.size *= 0.5f;
        out.kp.push_back(k);
    }
}

}  // namespace

extern "C" {

void so_sift_config_default(so_sift_config* c) { c->nfeatures = 0; c->n_octave_layers = 3; c->contrast_threshold = 0.04; c->edge_threshold = 10; c->sigma = 1.6; }

// returns the number of keypoints (writes at most cap); stats3 (may be null): octaves, raw extrema, refined extrema
int so_sift_bgr8(const uint8_t* bgr, int w, int h, int stride, const so_sift_config* sc, const slideo_ocv_variants* ocv,
                 slideo_keypoint* kp, uint8_t* desc128, int cap, int32_t* stats3) {
    SiftResult r;
    sift_detect_describe(bgr, w, h, stride, *sc, *ocv, r);
    const int n = (int)r.kp.size(), m = std::min(n, cap);
    if (kp) std::memcpy(kp, r.kp.data(), (size_t)m * sizeof(slideo_keypoint));
    if (desc128) std::memcpy(desc128, r.desc.data(), (size_t)m * 128);
    if (stats3) { stats3[0] = r.n_octaves; stats3[1] = r.n_extrema; stats3[2] = r.n_refined; }
    return n;
}

// pyramid tap: Gaussian layer (dog == 0) or DoG layer (dog != 0) `layer` of octave `octave`
int so_sift_layer(const uint8_t* bgr, int w, int h, int stride, const so_sift_config* sc, const slideo_ocv_variants* ocv, int octave, int layer, int dog,
                  float* out, int64_t cap, int32_t* lw, int32_t* lh) {
    SiftPyr P;
    sift_build(bgr, w, h, stride, *sc, *ocv, P);
    if (octave < 0 || octave >= P.n_oct || layer < 0 || layer >= (dog ? P.nl + 2 : P.nl + 3)) return 1;
    const ImgF& im = dog ? P.dog[(size_t)octave * (P.nl + 2) + layer] : P.g[(size_t)octave * (P.nl + 3) + layer];
    *lw = im.w; *lh = im.h;
    if ((int64_t)im.d.size() > cap) return 7;
    std::memcpy(out, im.d.data(), im.d.size() * 4);
    return 0;
}

}  // extern "C"
