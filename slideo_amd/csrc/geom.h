// geom.h — host-side geometry and constant tables for the ORB / verify kernels.
//
// Everything here is a pure function of the config and the image size and is
// computed once per (matcher, frame size) on the host in the same double /
// float arithmetic OpenCV 4.5.2 uses, then uploaded.
// Shared POD structs below are passed to kernels by value.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

#include "slideo_amd.h"

namespace slideo {

constexpr int MAX_LEVELS = 16;
constexpr int MAX_DIM = 4096;            // x,y packed in 12 bits each
constexpr int FAST_TW = 126, FAST_TH = 32; // FAST tile (outputs); + 1-px score halo = 128 x 34 score positions
constexpr int BLUR_RH = 56;                // blur: output rows per wave (multiple of 7: the register ring of blur_kernel; and of 8: four row pairs per round of blur_f32_kernel)
constexpr int BLUR_TW = 248, BLUR_TH = 4 * BLUR_RH;  // blur tile (outputs) per 256-thread block: 62 inner lanes x 4 px, 4 waves stacked
constexpr int KP_SORT_LDS = 8192;          // items the in-LDS canonical sort holds (64 KB); larger frames sort in global memory
constexpr int RNG_TABLE_MIN = 16384;       // pre-drawn cv::RNG outputs for RANSAC: at least this many, 4 per iteration + slack; grown on demand

struct LevelGeom {
    int32_t w, h, pitch;       // level image
    int32_t quota;             // retainBest n for this level
    int64_t ofs;               // byte offset of the level inside one frame's pyramid
    int32_t rx0, ry0, rx1, ry1;  // FAST keep-region [rx0,rx1) x [ry0,ry1)  (empty if rx1<=rx0)
    int32_t ftx, fty, ftile0;  // FAST tiles in x / y, first tile id of this level
    int32_t btx, bty, btile0;  // blur tiles
    int32_t cand_ofs, cand_cap;  // candidate list slice (entries) inside one frame's list
    int32_t xtab_ofs, ytab_ofs;  // resize coefficient tables (entries), valid for level >= 1
    float scale;               // (float)pow(scale_factor, level)
    int32_t xctab_ofs;         // packed x weights (256-c1) | c1 << 16, same indexing as xtab_ofs
    // resize_quad_kernel's tables (per group of 4 output columns; valid when rq_ok): the first source byte of the group's 8-byte
    // window, and one v_perm_b32 selector per output (bytes (p, 0, p + 1, 0), p = the output's left tap inside the window)
    int32_t rq_ok, xs_ofs, xsel_ofs;
};

struct PyrGeom {
    int32_t nlevels, w, h;
    int32_t fast_tiles, blur_tiles;
    int32_t cand_per_frame;       // entries
    int64_t frame_bytes;          // pyramid bytes per frame (multiple of 256)
    int32_t edge, fast_thr, half_patch, patch_size;
    LevelGeom lv[MAX_LEVELS];
};

inline int cv_round_d(double v) { return (int)std::lrint(v); }

// [OCV A.2] level sizes and scales (orb.cpp getScale / detectAndCompute)
inline void pyramid_dims(int w, int h, const slideo_config& c, int* ws, int* hs, float* sc) {
    for (int l = 0; l < c.nlevels; ++l) {
        float s = (float)std::pow((double)c.scale_factor, (double)l);
        sc[l] = s;
        ws[l] = cv_round_d((double)((float)w / s));
        hs[l] = cv_round_d((double)((float)h / s));
    }
}

// [OCV A.4] per-level feature quotas (orb.cpp computeKeyPoints)
inline void level_quotas(const slideo_config& c, int* q) {
    float factor = (float)(1.0 / (double)c.scale_factor);
    float nd = (float)c.nfeatures * (1.0f - factor) / (1.0f - (float)std::pow((double)factor, (double)c.nlevels));
    int sum = 0;
    for (int l = 0; l < c.nlevels - 1; ++l) {
        q[l] = cv_round_d((double)nd);
        sum += q[l];
        nd *= factor;
    }
    q[c.nlevels - 1] = std::max(c.nfeatures - sum, 0);
}

// [OCV A.5] umax (orb.cpp)
inline void umax_table(int half, int* umax /* half+2 */) {
    for (int i = 0; i < half + 2; ++i) umax[i] = 0;
    int vmax = (int)std::floor(half * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(half * std::sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = cv_round_d(std::sqrt((double)half * half - (double)v * v));
    for (int v = half, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// Weights of the intensity-centroid disc for describe_kernel: entry (row v + half, dword column c) -> two dwords of
// four byte weights each: (u + half) inside the disc else 0, and 1 inside the disc else 0, for u = 4c + j - half.
// `shift` = log2 of the padded dword columns per row; the table is zero-padded to a multiple of 64 entries.
inline void ic_weight_table(int half, const int* umax, std::vector<uint32_t>& tab, int& shift) {
    const int n = 2 * half + 1, ncol = (n + 3) / 4;
    shift = 0;
    while ((1 << shift) < ncol) ++shift;
    const int ncp = 1 << shift;
    size_t entries = ((size_t)n * ncp + 63) & ~(size_t)63;
    tab.assign(entries * 2, 0u);
    for (int r = 0; r < n; ++r) {
        const int v = r - half, av = v < 0 ? -v : v;
        for (int c = 0; c < ncol; ++c) {
            uint32_t wu = 0, wm = 0;
            for (int j = 0; j < 4; ++j) {
                const int uu = 4 * c + j, u = uu - half, au = u < 0 ? -u : u;
                if (uu < n && au <= umax[av]) { wu |= (uint32_t)uu << (8 * j); wm |= 1u << (8 * j); }
            }
            tab[((size_t)r * ncp + c) * 2] = wu;
            tab[((size_t)r * ncp + c) * 2 + 1] = wm;
        }
    }
}

// [OCV A.7] cv::RNG (64-bit multiply-with-carry)
struct CvRng {
    uint64_t state, mul;            // mul = CV_RNG_COEFF (slideo_ocv_variants.rng_mul, 4164903690)
    explicit CvRng(uint64_t s, uint32_t mul_ = 4164903690u) : state(s ? s : 0xffffffffULL), mul(mul_) {}
    uint32_t next() {
        state = (uint64_t)(uint32_t)state * mul + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a)) + a; }
};

// [OCV A.7] makeRandomPattern(patchSize, 512)
inline void brief_pattern(int patch_size, int8_t* xy /* 1024 */, uint32_t rng_mul = 4164903690u) {
    CvRng rng(0x34985739, rng_mul);
    for (int i = 0; i < 512; ++i) {
        xy[2 * i] = (int8_t)rng.uniform(-patch_size / 2, patch_size / 2 + 1);
        xy[2 * i + 1] = (int8_t)rng.uniform(-patch_size / 2, patch_size / 2 + 1);
    }
}

// [OCV A.6] the 7-tap sigma-2 Gaussian of ORB's blur, one form per slideo_ocv_variants.blur value:
//   0 / 1  f32 taps of getGaussianKernel(7, 2, CV_32F) (sepFilter2D in f32; gauss7_f32)
//   2      Q8 taps cvRound(k * 256) of sepFilter2D's integer path before OpenCV 4.2: 18 34 49 55 49 34 18 (sum 257)
//   3      error-diffused Q8 taps of GaussianBlur's bit-exact fixed-point path: 18 34 48 56 48 34 18 (sum 256)
inline void gauss7_f32(float* k) {
    double v[3], sum = 0;
    for (int i = 0, x = -6; i < 3; ++i, x += 2) { v[i] = std::exp((double)(x * x) * (-0.125 / 4.0)); sum += v[i]; }
    sum *= 2.0; sum += 1.0;
    const double mul1 = 1.0 / sum;
    for (int i = 0; i < 3; ++i) { k[i] = (float)(v[i] * mul1); k[6 - i] = k[i]; }
    k[3] = (float)(1.0 * mul1);
}
inline void gauss7_q8_rounded(int* k) {
    float kf[7];
    gauss7_f32(kf);
    for (int i = 0; i < 7; ++i) k[i] = (int)std::lrint((double)kf[i] * 256.0);
}
inline void gauss7_fixed(int* k) {
    double kd[7], sum = 0;
    for (int i = 0; i < 7; ++i) { double x = i - 3.0; kd[i] = std::exp(-0.5 / 4.0 * x * x); sum += kd[i]; }
    for (int i = 0; i < 7; ++i) kd[i] *= 1.0 / sum;
    double err = 0; int acc = 0;
    for (int i = 0; i < 3; ++i) {
        double adj = kd[i] * 256.0 + err;
        int v0 = cv_round_d(adj);
        err = adj - (double)v0;
        k[i] = k[6 - i] = v0;
        acc += v0;
    }
    k[3] = 256 - 2 * acc;
}

// [OCV A.2] INTER_LINEAR_EXACT per-axis coefficients: entry = ofs | c1 << 16
// (value = (256 - c1) * src[ofs] + c1 * src[min(ofs+1, n-1)])
// round_variant = slideo_ocv_variants.resize: 0 = cvRound (ties to even), 1 = floor(x + 0.5)
inline void linear_exact_table(int ssize, int dsize, std::vector<uint32_t>& out, int round_variant = 0) {
    double inv_scale = (double)dsize / (double)ssize, scale = 1.0 / inv_scale;
    for (int d = 0; d < dsize; ++d) {
        double f = scale * ((double)d + 0.5) - 0.5;
        int i = (int)std::floor(f);
        uint32_t ofs, c1;
        if (i >= 0 && ssize > 1) {
            if (i < ssize - 1) { ofs = (uint32_t)i; c1 = round_variant == 1 ? (uint32_t)std::floor((f - (double)i) * 256.0 + 0.5) : (uint32_t)cv_round_d((f - (double)i) * 256.0); }
            else { ofs = (uint32_t)(ssize - 1); c1 = 0; }
        } else { ofs = 0; c1 = 0; }
        out.push_back(ofs | (c1 << 16));
    }
}

inline bool config_supported(const slideo_config& c, const char** why) {
    int half = c.patch_size / 2;
    int desc_r = (int)std::ceil(half * std::sqrt(2.0));
    *why = "";
    if (c.nlevels < 1 || c.nlevels > MAX_LEVELS) { *why = "nlevels must be 1..16"; return false; }
    if (c.nfeatures < 1) { *why = "nfeatures must be >= 1"; return false; }
    if (c.patch_size == 31) { *why = "patch_size 31 selects OpenCV's learned pattern, which is not restated; the reference uses 62"; return false; }
    if (c.patch_size < 2 || half > 63) { *why = "patch_size out of range"; return false; }
    if (c.edge_threshold < desc_r + 3 || c.edge_threshold < half || c.edge_threshold < 4) {
        *why = "edge_threshold must cover the rotated BRIEF radius + blur support"; return false;
    }
    if (!(c.scale_factor > 1.0f)) { *why = "scale_factor must be > 1"; return false; }
    if (c.fast_threshold < 1 || c.fast_threshold > 254) { *why = "fast_threshold out of range"; return false; }
    if (c.knn_k < 1 || c.knn_k > 32) { *why = "knn_k must be 1..32"; return false; }
    if (!(c.ratio_test >= 0.f) || (c.ratio_test > 0.f && c.knn_k < 2)) { *why = "ratio_test must be >= 0 and needs knn_k >= 2"; return false; }
    if (c.max_candidate_pages < 1 || c.max_candidate_pages > 64) { *why = "max_candidate_pages must be 1..64"; return false; }
    if (c.max_rated < 1 || c.max_rated > 16) { *why = "max_rated must be 1..16"; return false; }
    if (c.ransac_max_iters < 1 || c.ransac_max_iters > 1000000) { *why = "ransac_max_iters must be 1..1000000"; return false; }
    if (c.small_area < 64) { *why = "small_area too small"; return false; }
    // OpenCV-variant switches: the values this library implements ([hip] in slideo_amd.h)
    const slideo_ocv_variants& o = c.ocv;
    if (o.gray < 0 || o.gray > 1) { *why = "ocv.gray must be 0 or 1"; return false; }
    if (o.blur < 0 || o.blur > 3) { *why = "ocv.blur must be 0..3"; return false; }
    if (o.resize < 0 || o.resize > 1) { *why = "ocv.resize must be 0 or 1"; return false; }
    if (o.atan < 0 || o.atan > 1) { *why = "ocv.atan must be 0 or 1"; return false; }
    if (o.area < 0 || o.area > 1) { *why = "ocv.area must be 0 or 1"; return false; }
    if (o.warp != 0) { *why = "ocv.warp: only 0 (10-bit fixed point) is implemented on the GPU; the CPU restatement has 1"; return false; }
    if (o.lm != 0) { *why = "ocv.lm: only 0 (Gaussian elimination) is implemented on the GPU; the CPU restatement has 1"; return false; }
    if (o.hdlt < 0 || o.hdlt > 2) { *why = "ocv.hdlt must be 0, 1 or 2"; return false; }
    if (c.verify_model < 0 || c.verify_model > 1) { *why = "verify_model must be 0 (similarity) or 1 (homography)"; return false; }
    if (c.matcher < 0 || c.matcher > 1) { *why = "matcher must be 0 (exact) or 1 (LSH-compatible)"; return false; }
    if (c.verdict_rule < 0 || c.verdict_rule > 1) { *why = "verdict_rule must be 0 (best similarity, the reference) or 1 (rating order, the similarity only accepts)"; return false; }
    if (c.matcher == 1 && (c.lsh_tables < 1 || c.lsh_tables > 8 || c.lsh_key_bits < 1 || c.lsh_key_bits > 16 || c.lsh_multi_probe < 0 || c.lsh_multi_probe > 2)) {
        *why = "lsh_tables must be 1..8, lsh_key_bits 1..16, lsh_multi_probe 0..2"; return false;
    }
    if (c.matcher == 1 && c.ratio_test > 0.f) { *why = "the ratio test needs the exact two nearest rows: matcher 0"; return false; }
    return true;
}

// ---- slideo_config.matcher 1: FLANN LshIndex's tables (flann/lsh_table.h LshTable<unsigned char>::initialize, recalled) --------
// Table i is keyed by key_bits bit positions of the 256-bit descriptor: cv::randShuffle of the positions 0..255 (for i < 256:
// j = rng % 256, swap(a[j], a[i])) on the thread's default cv::RNG (state 0xffffffff), continuing from table to table, and the
// first key_bits positions of the shuffled array; a key packs those bits in ascending position order (getKey walks the mask).
constexpr int LSH_MAX_TABLES = 8, LSH_MAX_BITS = 16;
struct LshParams { int32_t ntab, kb, mp; int32_t bit[LSH_MAX_TABLES][LSH_MAX_BITS]; };

inline LshParams lsh_params(const slideo_config& c) {
    LshParams L{};
    L.ntab = c.lsh_tables; L.kb = c.lsh_key_bits; L.mp = c.lsh_multi_probe;
    CvRng rng(0xffffffffULL, c.ocv.rng_mul);
    for (int t = 0; t < L.ntab; ++t) {
        int a[256];
        for (int i = 0; i < 256; ++i) a[i] = i;
        for (int i = 0; i < 256; ++i) { const int j = (int)(rng.next() % 256u); std::swap(a[j], a[i]); }
        std::sort(a, a + L.kb);
        for (int b = 0; b < L.kb; ++b) L.bit[t][b] = a[b];
    }
    return L;
}
inline uint32_t lsh_key_host(const LshParams& L, int t, const uint8_t* desc) {
    uint32_t k = 0;
    for (int b = 0; b < L.kb; ++b) { const int p = L.bit[t][b]; k |= (uint32_t)((desc[p >> 3] >> (p & 7)) & 1) << b; }
    return k;
}

// Builds the pyramid geometry + resize tables for a w x h input.
inline void build_pyr_geom(int w, int h, const slideo_config& c, PyrGeom& g, std::vector<uint32_t>& lin_tab) {
    int ws[MAX_LEVELS], hs[MAX_LEVELS], quota[MAX_LEVELS];
    float sc[MAX_LEVELS];
    pyramid_dims(w, h, c, ws, hs, sc);
    level_quotas(c, quota);
    g = PyrGeom();
    g.nlevels = c.nlevels; g.w = w; g.h = h;
    g.edge = c.edge_threshold; g.fast_thr = c.fast_threshold;
    g.half_patch = c.patch_size / 2; g.patch_size = c.patch_size;
    lin_tab.clear();
    int64_t ofs = 0; int ftile = 0, btile = 0, cand = 0;
    for (int l = 0; l < c.nlevels; ++l) {
        LevelGeom& L = g.lv[l];
        L.w = std::max(ws[l], 0); L.h = std::max(hs[l], 0);
        L.pitch = (L.w + 15) & ~15;
        L.quota = quota[l]; L.scale = sc[l];
        L.ofs = ofs;
        ofs += ((int64_t)L.pitch * L.h + 255) & ~(int64_t)255;
        const int e = c.edge_threshold;
        if (L.w > 2 * e && L.h > 2 * e) { L.rx0 = e; L.ry0 = e; L.rx1 = L.w - e; L.ry1 = L.h - e; }
        else { L.rx0 = L.ry0 = L.rx1 = L.ry1 = 0; }
        int rw = L.rx1 - L.rx0, rh = L.ry1 - L.ry0;
        L.ftx = rw > 0 ? (rw + FAST_TW - 1) / FAST_TW : 0;
        L.fty = rh > 0 ? (rh + FAST_TH - 1) / FAST_TH : 0;
        L.ftile0 = ftile; ftile += L.ftx * L.fty;
        L.btx = L.w > 0 ? (L.w + BLUR_TW - 1) / BLUR_TW : 0;
        L.bty = L.h > 0 ? (L.h + BLUR_TH - 1) / BLUR_TH : 0;
        L.btile0 = btile; btile += L.btx * L.bty;
        // NMS survivors are strict 3x3 maxima: at most one per 2x2 block of the keep-region
        L.cand_ofs = cand;
        L.cand_cap = rw > 0 ? ((rw + 1) / 2) * ((rh + 1) / 2) : 0;
        cand += L.cand_cap;
        if (l >= 1 && L.w > 0 && L.h > 0 && g.lv[l - 1].w > 0) {
            // x tables: 16-byte aligned, padded to a multiple of 4 entries with copies of the last entry
            // (resize_kernel reads them 4 entries at a time)
            while (lin_tab.size() % 4) lin_tab.push_back(0);
            L.xtab_ofs = (int32_t)lin_tab.size(); linear_exact_table(g.lv[l - 1].w, L.w, lin_tab, c.ocv.resize);
            while (lin_tab.size() % 4) lin_tab.push_back(lin_tab.back());
            L.xctab_ofs = (int32_t)lin_tab.size();
            for (size_t i = (size_t)L.xtab_ofs, e = lin_tab.size(); i < e; ++i) {
                uint32_t c1 = lin_tab[i] >> 16;
                lin_tab.push_back((256u - c1) | (c1 << 16));
            }
            while (lin_tab.size() % 4) lin_tab.push_back(0);
            L.ytab_ofs = (int32_t)lin_tab.size(); linear_exact_table(g.lv[l - 1].h, L.h, lin_tab, c.ocv.resize);
            while (lin_tab.size() % 4) lin_tab.push_back(lin_tab.back());     // (resize_quad_kernel reads 4 rows' entries as one uint4)
            // quad tables: every group of 4 outputs must tap inside 8 consecutive source bytes (shrink factors < ~2.3; a right tap
            // at byte 8 is allowed when its weight is 0 — the selector then picks a constant byte)
            const int nxq = (L.w + 3) / 4, sp = g.lv[l - 1].pitch;
            std::vector<uint32_t> xs(nxq), sel((size_t)nxq * 4);
            bool ok = sp >= 8;
            for (int q = 0; q < nxq && ok; ++q) {
                const uint32_t* xe = &lin_tab[(size_t)L.xtab_ofs + 4 * q];
                const int x_s = std::min((int)(xe[0] & 0xffff), sp - 8);
                xs[q] = (uint32_t)x_s;
                for (int i = 0; i < 4; ++i) {
                    const int p = (int)(xe[i] & 0xffff) - x_s;
                    if (p < 0 || p > 7 || (p == 7 && (xe[i] >> 16) != 0)) ok = false;
                    sel[(size_t)4 * q + i] = (uint32_t)p * 0x00010001u + 0x0c010c00u;
                }
            }
            L.rq_ok = ok ? 1 : 0;
            if (ok) {
                L.xs_ofs = (int32_t)lin_tab.size(); lin_tab.insert(lin_tab.end(), xs.begin(), xs.end());
                while (lin_tab.size() % 4) lin_tab.push_back(0);
                L.xsel_ofs = (int32_t)lin_tab.size(); lin_tab.insert(lin_tab.end(), sel.begin(), sel.end());
            }
        }
    }
    g.frame_bytes = ofs; g.fast_tiles = ftile; g.blur_tiles = btile; g.cand_per_frame = cand;
}

// ---- INTER_AREA tap tables ([OCV A.11] computeResizeAreaTab) -----------------
struct AreaTap { int32_t si; float alpha; };

struct AreaGeom {        // one per distinct (source size -> small size) class
    int32_t sw, sh;      // source (page / frame) size
    int32_t dw, dh;      // small size
    int32_t fast;        // integer-scale fast path (ResizeAreaFast)
    int32_t iscale_x, iscale_y;
    float fast_scale;    // 1/(iscale_x*iscale_y)
    int32_t xtap_ofs, xidx_ofs;   // taps (AreaTap) and per-dx start indices (dw+1 ints), generic path
    int32_t ytap_ofs, yidx_ofs;
    int32_t max_xtaps, max_ytaps;
    // re-projection through a tile of the WARPED image (verify.hip.h reproject_vt_kernel): the largest source span of a 32 x 8
    // tile of small pixels, and whether the class fits that kernel's limits (else reproject_kernel's frame window does it)
    int32_t vt_ok, vt_spw, vt_sph;
    int32_t xrec_ofs, yrec_ofs;   // AreaRec per small column / row (vt_ok classes)
};

// One small pixel's taps along an axis as ONE 32-byte record (two 16-byte loads instead of the index pair, the first tap and
// up to 7 weights one by one): first source pixel | count << 24, then the weights, 0 past the count.
struct alignas(16) AreaRec { int32_t first_n; float alpha[7]; };

// limits of reproject_vt_kernel: a thread prefetches VT_CG x VT_RI source pixels of a tile (32-column groups x 8-row steps)
constexpr int AREA_TW = 32, AREA_TH = 8;       // small pixels per tile (= SM_TW x SM_TH of verify.hip.h)
constexpr int VT_CG = 5, VT_RI = 5, VT_PX = 6144;

// variant = slideo_ocv_variants.area: 0 = computeResizeAreaTab, 1 = exact box-overlap weights (no 1e-3 cut-off)
inline void area_taps(int ssize, int dsize, double scale, std::vector<AreaTap>& taps, std::vector<int32_t>& idx, int& max_taps, int variant = 0) {
    max_taps = 0;
    for (int dx = 0; dx < dsize; ++dx) {
        idx.push_back((int32_t)taps.size());
        size_t before = taps.size();
        if (variant == 1) {
            const double a = dx * scale, b = std::min(a + scale, (double)ssize), cellw = b - a;
            for (int sx = (int)std::floor(a); sx < ssize && (double)sx < b; ++sx) {
                const double ov = std::min(b, sx + 1.0) - std::max(a, (double)sx);
                if (ov > 0) taps.push_back({sx, (float)(ov / cellw)});
            }
            max_taps = std::max(max_taps, (int)(taps.size() - before));
            continue;
        }
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) taps.push_back({sx1 - 1, (float)((sx1 - fsx1) / cell)});
        for (int sx = sx1; sx < sx2; ++sx) taps.push_back({sx, (float)(1.0 / cell)});
        if (fsx2 - sx2 > 1e-3) taps.push_back({sx2, (float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)});
        max_taps = std::max(max_taps, (int)(taps.size() - before));
    }
    idx.push_back((int32_t)taps.size());
}

// to_small_image target size (crates/matching-opencv/src/image_utils.rs:8-16)
inline void small_size(int w, int h, int small_area, int& sw, int& sh) {
    float factor = std::sqrt((float)small_area / (float)(w * h));
    sw = (int)((float)w * factor);
    sh = (int)((float)h * factor);
}

// Returns false when the resize is not a shrink (INTER_AREA would fall back to bilinear).
inline bool build_area_geom(int w, int h, int small_area, AreaGeom& a, std::vector<AreaTap>& taps, std::vector<int32_t>& idx, int variant = 0,
                            std::vector<AreaRec>* recs = nullptr) {
    a = AreaGeom();
    a.sw = w; a.sh = h;
    small_size(w, h, small_area, a.dw, a.dh);
    if (a.dw <= 0 || a.dh <= 0 || a.dw > w || a.dh > h) return false;
    double scale_x = 1. / ((double)a.dw / w), scale_y = 1. / ((double)a.dh / h);
    a.iscale_x = (int)std::llrint(scale_x); a.iscale_y = (int)std::llrint(scale_y);
    a.fast = std::fabs(scale_x - a.iscale_x) < DBL_EPSILON && std::fabs(scale_y - a.iscale_y) < DBL_EPSILON;
    a.fast_scale = 1.f / (float)(a.iscale_x * a.iscale_y);
    a.xidx_ofs = (int32_t)idx.size(); a.xtap_ofs = (int32_t)taps.size();
    {
        std::vector<AreaTap> t; std::vector<int32_t> i; area_taps(w, a.dw, scale_x, t, i, a.max_xtaps, variant);
        taps.insert(taps.end(), t.begin(), t.end()); idx.insert(idx.end(), i.begin(), i.end());
    }
    a.yidx_ofs = (int32_t)idx.size(); a.ytap_ofs = (int32_t)taps.size();
    {
        std::vector<AreaTap> t; std::vector<int32_t> i; area_taps(h, a.dh, scale_y, t, i, a.max_ytaps, variant);
        taps.insert(taps.end(), t.begin(), t.end()); idx.insert(idx.end(), i.begin(), i.end());
    }
    // source spans of the tiles; the taps of an output are consecutive source pixels (computeResizeAreaTab: optional left
    // fraction, whole pixels, optional right fraction) — checked, reproject_vt_kernel addresses them as first + k
    auto span = [&](int tap_ofs, int idx_ofs, int dsize, int tile, bool& consecutive) {
        int mx = 0;
        for (int d0 = 0; d0 < dsize; d0 += tile) {
            const int d1 = std::min(dsize, d0 + tile) - 1;
            mx = std::max(mx, taps[tap_ofs + idx[idx_ofs + d1 + 1] - 1].si - taps[tap_ofs + idx[idx_ofs + d0]].si + 1);
        }
        for (int d = 0; d < dsize; ++d)
            for (int k = idx[idx_ofs + d] + 1; k < idx[idx_ofs + d + 1]; ++k) consecutive &= taps[tap_ofs + k].si == taps[tap_ofs + k - 1].si + 1;
        return mx;
    };
    bool cons = true;
    a.vt_spw = span(a.xtap_ofs, a.xidx_ofs, a.dw, AREA_TW, cons);
    a.vt_sph = span(a.ytap_ofs, a.yidx_ofs, a.dh, AREA_TH, cons);
    a.vt_ok = recs && !a.fast && cons && a.max_xtaps <= 7 && a.max_ytaps <= 7 && std::max(w, h) < (1 << 24) &&
              a.vt_spw <= 32 * VT_CG && a.vt_sph <= 8 * VT_RI && a.vt_spw * a.vt_sph <= VT_PX;
    if (a.vt_ok) {
        auto emit = [&](int tap_ofs, int idx_ofs, int dsize) {
            const int32_t at = (int32_t)recs->size();
            for (int d = 0; d < dsize; ++d) {
                const int b = idx[idx_ofs + d], e = idx[idx_ofs + d + 1];
                AreaRec r{};
                r.first_n = taps[tap_ofs + b].si | ((e - b) << 24);
                for (int k = 0; k < 7; ++k) r.alpha[k] = b + k < e ? taps[tap_ofs + b + k].alpha : 0.f;
                recs->push_back(r);
            }
            return at;
        };
        a.xrec_ofs = emit(a.xtap_ofs, a.xidx_ofs, a.dw);
        a.yrec_ofs = emit(a.ytap_ofs, a.yidx_ofs, a.dh);
    }
    return true;
}

}  // namespace slideo
