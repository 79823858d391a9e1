// sift.hip.h — SIFT keypoint detect + describe on gfx950 (BASELINE configs[2] / north_star "SIFT keypoint detect+describe";
// SURVEY 8(f) N4).  The reference has no SIFT (its only extractor is ORB, crates/matching-opencv/src/feature_extractor.rs:3-4,13):
// the parity target is the CPU restatement of cv::SIFT::detectAndCompute (OpenCV 4.5.2, recalled) in oracle/sift_oracle.h, whose
// header lists what is restated and the two documented departures (fixed-point histogram sums, canonical keypoint order).
// Every float operation below is written in the oracle's order, so the whole extractor is bit-exact against it: pyramid layers,
// keypoints and the 128 descriptor bytes.
//
//   sift_base_kernel      gray (u8, the ORB path's gray_kernel) -> f32, 2x bilinear upsample (firstOctave = -1)
//   sift_blur_kernel      GaussianBlur on f32 = sepFilter2D: row pass (taps in order) and column pass (centre, then symmetric
//                         pairs) through LDS in ONE launch per layer, BORDER_REFLECT_101; optionally writes the DoG layer
//                         (this layer minus its input) in the same pass — the difference pyramid costs no read of its own
//   sift_half_kernel      next octave's first layer: every second pixel of layer nOctaveLayers
//   sift_extrema_kernel   |v| > threshold and >= / <= all 26 neighbours -> candidate list (atomic append; order is restored later)
//   sift_refine_kernel    one wave per candidate: adjustLocalExtrema (every lane the same few dozen flops), then the
//                         orientation histogram — samples across the lanes, 36 bins of 64-bit fixed point in LDS (integer
//                         atomics: order independent) — smoothing, peaks -> raw keypoints
//   sift_select_kernel    one block per frame: canonical order (bitonic sort of keys in global memory), duplicates out,
//                         retainBest(nfeatures) by a 4-pass radix select on the response bits (ties kept)
//   sift_describe_kernel  one wave per kept keypoint: 4 x 4 x 8 histogram (360 fixed-point bins in LDS), trilinear votes across
//                         the lanes, normalisation and the 0.2 clamp in the oracle's (serial) summation order, 128 bytes out
//
// Bounds: HBM.  Per 1080p frame the doubled base image is 3840 x 2160 f32 = 33 MB and the pyramid writes (6 Gaussian + 5 DoG
// layers) x 4/3 octaves x 33 MB = 486 MB and reads about as much again: ~1 GB per frame, 120 us at 8 TB/s — SIFT with the
// doubled first octave is two orders of magnitude more traffic than ORB's byte pyramid (42 MB).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "geom.h"
#include "types.h"
#include "cv_math.hip.h"

namespace slideo {

constexpr int SIFT_MAX_OCT = 12;
constexpr int SIFT_NL = 3;                      // nOctaveLayers (the only value the kernels implement)
constexpr int SIFT_BORDER = 5, SIFT_STEPS = 5, SIFT_BINS = 36;
constexpr int SIFT_MAX_TAPS = 40;               // cvRound(8 s + 1) | 1 of the widest layer blur s = 1.93 sigma: 27 taps at sigma 1.6, 39 at 2.4
constexpr int SIFT_BT_W = 64, SIFT_BT_H = 32;   // blur tile (outputs) per 256-thread block
constexpr double SIFT_FIXD = 1048576.0;         // 2^20

struct SiftGeom {
    int32_t w, h;                               // input image
    int32_t n_oct;
    int32_t ow[SIFT_MAX_OCT], oh[SIFT_MAX_OCT];
    int64_t g_ofs[SIFT_MAX_OCT], d_ofs[SIFT_MAX_OCT];      // float offsets of an octave's first Gaussian / DoG layer inside one frame
    int64_t g_frame, d_frame;                   // floats per frame
};

struct SiftTaps { int32_t n; float k[SIFT_MAX_TAPS]; };

struct SiftParams {
    int32_t nfeatures, cand_cap, raw_cap;
    float contrast_threshold, edge_threshold, sigma;
    int32_t threshold;                          // floor(0.5 * contrastThreshold / nOctaveLayers * 255)
    int32_t atan_fma, blur_fma;
};

struct SiftRaw { slideo_keypoint kp; uint64_t key; };      // raw keypoint + canonical key (octave, layer, r, c, bin)

__device__ __forceinline__ int sift_reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

// The gray image first (gray_kernel, orb.hip.h: the ORB path's conversion, u8, pitch `gp`), then one thread per SOURCE pixel
// (i, j): the 2 x 2 block of outputs (2i .. 2i + 1, 2j .. 2j + 1) taps the 3 x 3 gray values around it — three unaligned 4-byte loads
// (reading the BGR frame directly cost 27 byte loads per thread: bound by the texture path at 30 us per 1080p frame).
// The interpolation is resize(INTER_LINEAR)'s: per output coordinate (i0, i1, a0, a1) from the same expression as the oracle
// (borders clamp to weight 1 / 0), t = g0 a0 + g1 a1 per row, then t0 b0 + t1 b1: unfused multiplies and adds.
// grid (ceil(w / 512), ceil(h / SIFT_BASE_ROWS), n); the gray rows must be readable 6 bytes past column w - 1 (a following row, or the slack stage_sift reserves).
constexpr int SIFT_BASE_ROWS = 4;          // source rows per thread (a thread per source pixel: 8.8 M waves of 30 instructions, 2.1 TB/s)
// One thread = source columns (i, i + 1), i even, x SIFT_BASE_ROWS source rows: one unaligned dword per gray row holds columns
// i - 1 .. i + 2, the outputs of a doubled row leave as one 16-byte store.
__global__ __launch_bounds__(256) void sift_base_kernel(const uint8_t* __restrict__ gray, int64_t gray_frame, int gp, int w, int h,
                                                        float* __restrict__ out, int64_t out_frame) {
    const int i = 2 * (blockIdx.x * 256 + threadIdx.x), j0 = blockIdx.y * SIFT_BASE_ROWS;
    if (i >= w) return;
    const int W = 2 * w;
    const bool two = i + 1 < w;                                              // (odd w: the last thread owns one column)
    const uint8_t* img = gray + (int64_t)blockIdx.z * gray_frame;
    // gray rows j0 - 1 .. j0 + SIFT_BASE_ROWS (clamped); per row the values at columns clamp(i - 1), i, clamp(i + 1), clamp(i + 2)
    float gr[SIFT_BASE_ROWS + 2][4];
    {
        const int xs = max(i - 1, 0);                                       // bytes xs .. xs + 3 = columns i - 1 .. i + 2 (i = 0: 0 .. 3)
#pragma unroll
        for (int q = 0; q < SIFT_BASE_ROWS + 2; ++q) {
            const int row = min(max(j0 - 1 + q, 0), h - 1);
            // (an aligned pair of dwords and a funnel shift: xs is odd for every thread but the first of a row, and the texture path
            // splits unaligned loads — resize_quad_kernel, orb.hip.h; up to 6 bytes past column w - 1 are read: stage_sift's slack)
            uint32_t w2[2];
            __builtin_memcpy(w2, img + (int64_t)row * gp + (xs & ~3), 8);
            const uint32_t v = __builtin_amdgcn_alignbit(w2[1], w2[0], (uint32_t)(xs & 3) * 8u);
            const float b0 = (float)(v & 255u), b1 = (float)((v >> 8) & 255u), b2 = (float)((v >> 16) & 255u), b3 = (float)(v >> 24);
            const float c0 = i >= 1 ? b1 : b0;                                            // column i
            const float c1 = i >= 1 ? b2 : b1;                                            // column i + 1 (if it exists)
            const float c2 = i >= 1 ? b3 : b2;                                            // column i + 2 (if it exists)
            gr[q][0] = b0;                                                                // column max(i - 1, 0)
            gr[q][1] = c0;
            gr[q][2] = i + 1 <= w - 1 ? c1 : c0;                                          // column min(i + 1, w - 1)
            gr[q][3] = i + 2 <= w - 1 ? c2 : (i + 1 <= w - 1 ? c1 : c0);                  // column min(i + 2, w - 1)
        }
    }
    // (index into the 3-neighbourhood of source coordinate q around centre `ctr`: 0, 1, 2 = ctr - 1, ctr, ctr + 1 after clamping)
    auto coef = [](int d, int n, int ctr, int& k0, int& k1, float& a0, float& a1) {
        float f = (float)(((double)d + 0.5) * 0.5 - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { s = 0; f = 0; }
        if (s >= n - 1) { s = n - 1; f = 0; }
        const int i0 = s, i1 = min(s + 1, n - 1);
        k0 = i0 < ctr ? 0 : (i0 > ctr ? 2 : 1); k1 = i1 < ctr ? 0 : (i1 > ctr ? 2 : 1);
        a0 = 1.f - f; a1 = f;
    };
    auto pick = [](const float (&row)[3], int k) -> float { return k == 0 ? row[0] : (k == 1 ? row[1] : row[2]); };
    float* o = out + (int64_t)blockIdx.z * out_frame + (int64_t)(2 * j0) * W + 2 * i;
#pragma unroll
    for (int u = 0; u < SIFT_BASE_ROWS; ++u) {
        const int j = j0 + u;
        if (j >= h) break;
        float res[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c) {                                       // source column ic = i + c: gray columns (ic - 1, ic, ic + 1) = gr[.][c .. c + 2]
            const int ic = i + c;
            float g[3][3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { g[q][0] = gr[u + q][c]; g[q][1] = gr[u + q][c + 1]; g[q][2] = gr[u + q][c + 2]; }
            if (ic >= 1 && ic <= w - 2 && j >= 1 && j <= h - 2) {
                // interior: the coefficient expression gives (ic - 1, ic; 0.25, 0.75) for output 2 ic and (ic, ic + 1; 0.75, 0.25) for
                // 2 ic + 1 — exactly (1 - 0.75 = 0.25 and 1 - 0.25 = 0.75 in f32) — and the same in y: no selection needed
                float ta[3], tb[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) { ta[q] = g[q][0] * 0.25f + g[q][1] * 0.75f; tb[q] = g[q][1] * 0.75f + g[q][2] * 0.25f; }
                res[0][2 * c] = ta[0] * 0.25f + ta[1] * 0.75f; res[0][2 * c + 1] = tb[0] * 0.25f + tb[1] * 0.75f;
                res[1][2 * c] = ta[1] * 0.75f + ta[2] * 0.25f; res[1][2 * c + 1] = tb[1] * 0.75f + tb[2] * 0.25f;
            } else
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                int ky0, ky1; float b0, b1;
                coef(2 * j + dy, h, j, ky0, ky1, b0, b1);
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    int kx0, kx1; float a0, a1;
                    coef(2 * ic + dx, w, min(ic, w - 1), kx0, kx1, a0, a1);
                    const float r0[3] = {g[0][0], g[1][0], g[2][0]}, r1[3] = {g[0][1], g[1][1], g[2][1]}, r2[3] = {g[0][2], g[1][2], g[2][2]};
                    // gray value at (row index ky, column index kx)
                    auto at = [&](int ky, int kx) -> float { return kx == 0 ? pick(r0, ky) : (kx == 1 ? pick(r1, ky) : pick(r2, ky)); };
                    const float t0 = at(ky0, kx0) * a0 + at(ky0, kx1) * a1;
                    const float t1 = at(ky1, kx0) * a0 + at(ky1, kx1) * a1;
                    res[dy][2 * c + dx] = t0 * b0 + t1 * b1;
                }
            }
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            float* d = o + (int64_t)(2 * u + dy) * W;
            if (two) *reinterpret_cast<float4*>(d) = make_float4(res[dy][0], res[dy][1], res[dy][2], res[dy][3]);
            else *reinterpret_cast<float2*>(d) = make_float2(res[dy][0], res[dy][1]);
        }
    }
}

// grid (tiles_x * tiles_y, n), block 256.  src / dst: layer pointers of frame 0, per-frame strides in floats.
// dog != null: dog = dst - src (the difference layer between the input and the output layer).
template <bool FMA>
__global__ __launch_bounds__(256) void sift_blur_kernel(const float* __restrict__ src, int64_t src_frame, float* __restrict__ dst, int64_t dst_frame,
                                                        float* __restrict__ dog, int64_t dog_frame, int w, int h, SiftTaps tp) {
    constexpr int TW = SIFT_BT_W, TH = SIFT_BT_H, MAXR = SIFT_MAX_TAPS / 2;
    __shared__ float s_in[(TH + 2 * MAXR) * (TW + 2 * MAXR)];
    __shared__ float s_row[(TH + 2 * MAXR) * TW];
    const int r = tp.n / 2;
    const int tiles_x = (w + TW - 1) / TW;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const float* S = src + (int64_t)blockIdx.y * src_frame;
    const int iw = TW + 2 * r, ih = TH + 2 * r;
    for (int i = threadIdx.x; i < iw * ih; i += 256) {
        const int yy = i / iw, xx = i - yy * iw;
        s_in[i] = S[(int64_t)sift_reflect101(y0 - r + yy, h) * w + sift_reflect101(x0 - r + xx, w)];
    }
    __syncthreads();
    auto mad = [](float a, float b, float c) -> float { return FMA ? __builtin_fmaf(a, b, c) : a * b + c; };
    // row pass: s = k[0] S[0], then += k[j] S[j] in tap order
    for (int i = threadIdx.x; i < ih * TW; i += 256) {
        const int yy = i / TW, xx = i - yy * TW;
        const float* p = s_in + yy * iw + xx;
        float s = tp.k[0] * p[0];
        for (int j = 1; j < tp.n; ++j) s = mad(tp.k[j], p[j], s);
        s_row[i] = s;
    }
    __syncthreads();
    // column pass: s = k[r] T[c], then += k[r + j] (T[c + j] + T[c - j])
    float* D = dst + (int64_t)blockIdx.y * dst_frame;
    for (int i = threadIdx.x; i < TH * TW; i += 256) {
        const int yy = i / TW, xx = i - yy * TW;
        const int gx = x0 + xx, gy = y0 + yy;
        if (gx >= w || gy >= h) continue;
        const float* p = s_row + (yy + r) * TW + xx;
        float s = tp.k[r] * p[0];
        for (int j = 1; j <= r; ++j) s = mad(tp.k[r + j], p[j * TW] + p[-j * TW], s);
        D[(int64_t)gy * w + gx] = s;
        if (dog) dog[(int64_t)blockIdx.y * dog_frame + (int64_t)gy * w + gx] = s - s_in[(yy + r) * iw + xx + r];
    }
}

// LDS accesses the compiler does not see.  It orders every LDS access it knows of after every LDS-DMA in flight (s_waitcnt
// vmcnt(0): it cannot tell the buffers apart), which would serialise a prefetch with the step that issues it — the stream
// kernel below waits for exactly the DMA it needs, by count, and touches the LDS only through these.
typedef float sift_f4 __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ sift_f4 sift_lds_rd128(uint32_t a) { sift_f4 r; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF)); return r; }
template <int OFF> __device__ __forceinline__ float sift_lds_rd32(uint32_t a) { float r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF)); return r; }
template <int OFF> __device__ __forceinline__ void sift_lds_wr128(uint32_t a, sift_f4 v) { asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(a), "v"(v), "n"(OFF) : "memory"); }
__device__ __forceinline__ uint32_t sift_lds_addr(const void* p) { return (uint32_t)(size_t)(__attribute__((address_space(3))) const void*)p; }

// The same filter as a STREAM (the default for compile-time tap counts): one wave owns a strip of 64 output columns and walks
// down `chunk_h` rows of it, eight rows per step.  The tile kernel above pays for its two block barriers and for 58 KB of LDS
// (two blocks per CU: the fill of one tile is exposed while the other computes) and recomputes the row pass on a (64 + 2R) / 64
// vertical halo.  Here, per step of 8 rows:
//   * the 8 input rows (64 + 2R columns, reflected at the image borders element by element) of the NEXT step are requested by
//     LDS-DMA into one of three buffers (12 instructions of 64 dwords, row pitch 96) — no registers, two steps ahead;
//   * row pass: lane (row r of the 8, segment s of 8) reads the N + 7 floats its 8 outputs need from the buffer (ds_read_b128),
//     8 N fused multiply-adds in tap order, and hands the 8 results to the column lanes through a 2 KB staging tile;
//   * column pass — lane = column — keeps the last 8 (LAG + 1) row-pass values of its column in REGISTERS (a window shifted by 8
//     per step) and produces 8 vertically adjacent outputs (centre, then the symmetric pairs outward), LAG = ceil(2R / 8) steps
//     behind the row pass; writes the layer and, if asked, the difference to the input layer (one more coalesced read).
// A row-pass row is computed once per chunk (+ 2R warm-up rows), LDS is 11.4 KB per wave whatever N, no block barrier exists: the
// four waves of a block are independent strips.  Per output value the operation order is the tile kernel's and the generic
// kernel's: bit-identical.   grid (ceil(strips * chunks / 4), n), block 256.
template <int N, bool FMA, bool HALF = false>      // HALF: also write the every-second-pixel copy (instantiated for the layer-3 tap count only: the other instances keep their registers)
__global__ __attribute__((amdgpu_waves_per_eu(3))) __launch_bounds__(256) void sift_blur_stream_kernel(const float* __restrict__ src, int64_t src_frame, float* __restrict__ dst, int64_t dst_frame,
                                                               float* __restrict__ dog, int64_t dog_frame, int w, int h, SiftTaps tp, int chunk_h,
                                                               float* __restrict__ half, int64_t half_frame, int hw, int hh) {
    constexpr int R = N / 2, LAG = (2 * R + 7) / 8, WIN = 8 * (LAG + 1), IW = 64 + 2 * R, IP = 96, NV4 = (N + 7 + 3) / 4, SP = 68;
    // input buffers: three (rows requested two steps ahead) for the wide kernels, which are bound by their arithmetic and by
    // registers (3 waves per SIMD); two for the narrow ones — 33 instead of 45 KB of LDS per block: 4 instead of 3 waves per SIMD
    constexpr int NB = N >= 21 ? 3 : 2;
    static_assert(IW <= IP && 4 * NV4 + 56 <= IP, "input row pitch");
    __shared__ __attribute__((aligned(16))) float s_in[4][NB][8 * IP];
    __shared__ __attribute__((aligned(16))) float s_st[4][8 * SP];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strips_x = (w + 63) >> 6, chunks = (h + chunk_h - 1) / chunk_h;
    const int task = blockIdx.x * 4 + wave;
    if (task >= strips_x * chunks) return;                                  // (wave-uniform)
    const int cy = task / strips_x, sx = task - cy * strips_x;
    const int x0 = sx * 64, y0 = cy * chunk_h, y1 = min(h, y0 + chunk_h);
    const float* S = src + (int64_t)blockIdx.y * src_frame;
    float* D = dst + (int64_t)blockIdx.y * dst_frame;
    float* G = dog ? dog + (int64_t)blockIdx.y * dog_frame : nullptr;
    // `half` (the launch that completes layer nOctaveLayers): the next octave's first layer = every second pixel of this output,
    // written from the registers instead of being read back by sift_half_kernel (chunks start at even rows: chunk_h is a multiple of 8)
    float* HF = HALF && half ? half + (int64_t)blockIdx.y * half_frame : nullptr;
    float* const IN = s_in[wave][0];
    const uint32_t in_addr = sift_lds_addr(IN), st_addr = sift_lds_addr(s_st[wave]);
    // (the taps are symmetric bit for bit — exp(-x^2 / 2 sigma^2) of +-x — so R + 1 registers hold them)
    float kk[R + 1];
#pragma unroll
    for (int j = 0; j <= R; ++j) { kk[j] = tp.k[j]; asm volatile("" : "+v"(kk[j])); }
    auto K = [&](int j) -> float { return kk[j <= R ? j : N - 1 - j]; };
    auto mad = [](float a, float b, float c) -> float { return FMA ? __builtin_fmaf(a, b, c) : a * b + c; };
    // DMA map: 8 rows x 96 dwords = 12 instructions of 64 dwords; instruction 3m + u fills dwords 64 (3m + u) .. + 63 = rows 2m, 2m + 1:
    //   u = 0: row 2m, columns 0..63;  u = 1: lanes 0..31 row 2m columns 64..95, lanes 32..63 row 2m + 1 columns 0..31;  u = 2: row 2m + 1, columns 32..95
    int colofs[3]; bool second[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int d = 64 * u + lane, col = d >= IP ? d - IP : d;
        second[u] = d >= IP;
        colofs[u] = sift_reflect101(x0 - R + min(col, IW - 1), w);
    }
    // rows -R .. h + 8 LAG + 6: one reflection is enough for h >= 48 (a while loop per row otherwise: branches in every step)
    auto refl1 = [&](int p) -> int { return h >= 48 ? (p < 0 ? -p : (p >= h ? 2 * (h - 1) - p : p)) : sift_reflect101(p, h); };
    auto dma_rows = [&](int tb, int buf) {                                   // input rows tb .. tb + 7 -> IN[buf]
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int ra = refl1(tb + 2 * m), rb = refl1(tb + 2 * m + 1);          // (scalar)
            const float* pa = S + (int64_t)ra * w;
            const float* pb = S + (int64_t)rb * w;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const float* ga = (second[u] ? pb : pa) + colofs[u];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                                 (__attribute__((address_space(3))) void*)(IN + buf * (8 * IP) + 64 * (3 * m + u)), 4, 0, 0);
            }
        }
    };
    const int rr = lane >> 3, sg = lane & 7;
    const int gx = x0 + lane;
    const int nY = (y1 - y0 + 7) >> 3, M = nY + LAG;
    float c[WIN];
#pragma unroll
    for (int t = 0; t < WIN; ++t) c[t] = 0.f;
    // Vector-memory operations of a wave complete in order, so "at most 8 outstanding" right after [12 DMA loads][8 stores] means the
    // loads are in: a step's input wait then does not sit through the write acknowledgements of the previous step's outputs
    // (waiting for everything, the waves spent half their time there).  `stores8`: the previous step issued exactly that sequence.
    // The input rows are requested TWO steps ahead (three buffers): at the top of step m the operations issued after DMA(m) are, in
    // order, [stores of step m - 2][DMA(m + 1)][stores of step m - 1], so "at most s2 + 12 + s1 outstanding" means DMA(m) is in
    // (s = 8 for a step that issued exactly its 8 stores, else 0: a lower bound is always safe).
    int s1 = 0, s2 = 0, buf = 0;
    dma_rows(y0 - R, 0);
    if (NB == 3 && M > 1) dma_rows(y0 - R + 8, 1);
    for (int m = 0; m < M; ++m) {
        {
            // (two buffers: DMA(m + 1) is issued in step m, so only the stores of step m - 1 follow DMA(m))
            const int allow = NB == 3 ? s2 + s1 + (m + 1 < M ? 12 : 0) : s1;
            if (allow >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
            else if (allow >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else if (allow >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (allow >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s2 = s1; s1 = 0;
        __builtin_amdgcn_wave_barrier();
        {   // row pass
            float v[4 * NV4];
            {
                const uint32_t a = in_addr + (uint32_t)(buf * (8 * IP) + rr * IP + 8 * sg) * 4u;
                sift_f4 t4[NV4];
                static_assert(NV4 <= 9, "row-pass reads");
                t4[0] = sift_lds_rd128<0>(a);
                if constexpr (NV4 > 1) t4[1] = sift_lds_rd128<16>(a);
                if constexpr (NV4 > 2) t4[2] = sift_lds_rd128<32>(a);
                if constexpr (NV4 > 3) t4[3] = sift_lds_rd128<48>(a);
                if constexpr (NV4 > 4) t4[4] = sift_lds_rd128<64>(a);
                if constexpr (NV4 > 5) t4[5] = sift_lds_rd128<80>(a);
                if constexpr (NV4 > 6) t4[6] = sift_lds_rd128<96>(a);
                if constexpr (NV4 > 7) t4[7] = sift_lds_rd128<112>(a);
                if constexpr (NV4 > 8) t4[8] = sift_lds_rd128<128>(a);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < NV4; ++q) { asm volatile("" : "+v"(t4[q])); v[4 * q] = t4[q].x; v[4 * q + 1] = t4[q].y; v[4 * q + 2] = t4[q].z; v[4 * q + 3] = t4[q].w; }
            }
            float acc[8];
#pragma unroll
            for (int t = 0; t < N + 7; ++t) {
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const int j = t - o;
                    if (j == 0) acc[o] = K(0) * v[t];
                    else if (j > 0 && j < N) acc[o] = mad(K(j), v[t], acc[o]);
                }
            }
            const uint32_t a = st_addr + (uint32_t)(rr * SP + 8 * sg) * 4u;
            sift_lds_wr128<0>(a, sift_f4{acc[0], acc[1], acc[2], acc[3]});
            sift_lds_wr128<16>(a, sift_f4{acc[4], acc[5], acc[6], acc[7]});
        }
        __builtin_amdgcn_wave_barrier();
        // window: c[u] = row-pass value of row y0 - R + 8 (m - LAG) + u   (a wave's LDS operations execute in order: the reads
        // below see the writes above)
#pragma unroll
        for (int t = 0; t < WIN - 8; ++t) c[t] = c[t + 8];
        {
            const uint32_t a = st_addr + (uint32_t)lane * 4u;
            float nv[8];
            nv[0] = sift_lds_rd32<0>(a); nv[1] = sift_lds_rd32<SP * 4>(a); nv[2] = sift_lds_rd32<2 * SP * 4>(a); nv[3] = sift_lds_rd32<3 * SP * 4>(a);
            nv[4] = sift_lds_rd32<4 * SP * 4>(a); nv[5] = sift_lds_rd32<5 * SP * 4>(a); nv[6] = sift_lds_rd32<6 * SP * 4>(a); nv[7] = sift_lds_rd32<7 * SP * 4>(a);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 8; ++t) { asm volatile("" : "+v"(nv[t])); c[WIN - 8 + t] = nv[t]; }
        }
        __builtin_amdgcn_wave_barrier();
        // (the buffer requested into was last read by the previous step's row pass — three buffers — or by this step's — two)
        if (m + NB - 1 < M) dma_rows(y0 - R + 8 * (m + NB - 1), NB == 3 ? (buf == 0 ? 2 : buf - 1) : buf ^ 1);
        if (m >= LAG && !G && y0 + 8 * (m - LAG) + 8 <= y1) {
            // a full block of 8 output rows, layer only: exactly 8 store instructions after the 12 DMA loads
            const int Y = y0 + 8 * (m - LAG);
            float res[8];
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float sacc = K(R) * c[o + R];
#pragma unroll
                for (int j = 1; j <= R; ++j) sacc = mad(K(R + j), c[o + R + j] + c[o + R - j], sacc);
                res[o] = sacc;
            }
            if (gx < w) {
                float* o0 = D + (int64_t)Y * w + gx;
#pragma unroll
                for (int o = 0; o < 8; ++o) __builtin_nontemporal_store(res[o], o0 + (int64_t)o * w);
            }
            if (HF && !(gx & 1) && (gx >> 1) < hw) {
#pragma unroll
                for (int o = 0; o < 8; o += 2) if (((Y + o) >> 1) < hh) HF[(int64_t)((Y + o) >> 1) * hw + (gx >> 1)] = res[o];
            }
            s1 = 8;
        } else if (m >= LAG) {
            const int Y = y0 + 8 * (m - LAG);
            float sv[8];
            if (G) {
#pragma unroll
                for (int o = 0; o < 8; ++o) sv[o] = S[(int64_t)min(Y + o, h - 1) * w + min(gx, w - 1)];
            }
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float sacc = K(R) * c[o + R];
#pragma unroll
                for (int j = 1; j <= R; ++j) sacc = mad(K(R + j), c[o + R + j] + c[o + R - j], sacc);
                const int gy = Y + o;
                if (gx < w && gy < y1) {
                    D[(int64_t)gy * w + gx] = sacc;
                    if (G) G[(int64_t)gy * w + gx] = sacc - sv[o];
                    if (HF && !((gx | gy) & 1) && (gx >> 1) < hw && (gy >> 1) < hh) HF[(int64_t)(gy >> 1) * hw + (gx >> 1)] = sacc;
                }
            }
        }
        buf = buf == NB - 1 ? 0 : buf + 1;
    }
}

// dst(x, y) = src(2x, 2y).  grid (ceil(dw / 256), dh, n).  (Four outputs x four rows per thread through 16-byte accesses: measured
// 15 % slower — r03.)
__global__ __launch_bounds__(256) void sift_half_kernel(const float* __restrict__ src, int64_t src_frame, int sw, float* __restrict__ dst,
                                                        int64_t dst_frame, int dw, int dh) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    dst[(int64_t)blockIdx.z * dst_frame + (int64_t)y * dw + x] = src[(int64_t)blockIdx.z * src_frame + (int64_t)(2 * y) * sw + 2 * x];
}

// The difference-of-Gaussians pyramid is NOT stored: DoG layer L of an octave is Gaussian layer L + 1 minus layer L, one f32
// subtraction of two stored values — the same value whether the blur kernel writes it or a reader recomputes it (the blur
// kernels can still write it: `dog` argument).  Not storing it takes a third of the blur's HBM traffic and 5 / 11 of the workspace;
// the readers (extrema, refinement) load two values instead of one.
struct SiftDog {
    const float* g; int64_t lsz;                                            // Gaussian layer L (layer L + 1 is lsz floats further)
    __device__ __forceinline__ float operator[](int64_t i) const { return g[i + lsz] - g[i]; }
    __device__ __forceinline__ SiftDog operator+(int64_t d) const { return SiftDog{g + d, lsz}; }
    __device__ __forceinline__ SiftDog operator-(int64_t d) const { return SiftDog{g - d, lsz}; }
};

// candidate = o << 28 | layer << 26 | r << 13 | c.
// A point is an extremum iff it is the maximum (or minimum) of the 27 values around it in (x, y, layer) — itself included, so
// "val >= all 26 neighbours" is "val >= max27".  The 3 x 3 x 3 maximum is separable: a thread owns one column and walks down the
// rows of ALL FIVE difference layers of the octave (six Gaussian layers: one load each per row, five subtractions), the horizontal 3-maximum / 3-minimum through its
// neighbours' registers (DPP wave shifts), the vertical one over a 3-row register ring, the layer one across the rings — every
// pyramid value is read from memory exactly once (the one-thread-per-point form re-read 27 values wherever a wave held a point above
// the threshold, i.e. almost everywhere on a text frame, and was bound by the texture path: 30 ms per 256 frames).
// A wave covers 128 columns (two per lane) and emits for all but its first and last; block = 4 waves = 4 column groups; grid
// (ceil(cols / (4 SIFT_EX_COLS)), ceil(rows / SIFT_EX_RCH), n), cols / rows = the octave's sides minus the 5-px borders.
constexpr int SIFT_EX_RCH = 34;                 // output rows per block (+ 2 halo rows = 36 row steps, a multiple of the 3-slot ring)
constexpr int SIFT_EX_COLS = 126;               // output columns per wave
constexpr int SIFT_EX_STAGE = 512;              // candidates a wave stages in LDS before its one list append

#ifdef SIFT_EX_BPERMUTE
__device__ __forceinline__ float sift_wave_shr1(float v) { return __shfl_up(v, 1); }
__device__ __forceinline__ float sift_wave_shl1(float v) { return __shfl_down(v, 1); }
#else
__device__ __forceinline__ float sift_wave_shr1(float v) {      // lane i <- lane i - 1 (lane 0 keeps its own)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float sift_wave_shl1(float v) {      // lane i <- lane i + 1 (lane 63 keeps its own)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}
#endif

__global__ __launch_bounds__(256) void sift_extrema_kernel(SiftGeom g, SiftParams sp, int o, const float* __restrict__ gauss,
                                                           uint32_t* __restrict__ cand, uint32_t* __restrict__ cand_count, uint32_t* __restrict__ flags) {
    const int f = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = g.ow[o], h = g.oh[o];
    // a lane owns TWO adjacent columns (8-byte loads: 512 B per wave and row); the wave's first and last column are halo
    const int cw0 = SIFT_BORDER - 1 + (blockIdx.x * 4 + wave) * SIFT_EX_COLS;
    if (cw0 + 1 >= w - SIFT_BORDER) return;                                        // (wave-uniform: no output column in this group)
    const int c = cw0 + 2 * lane;
    const int ry0 = SIFT_BORDER + blockIdx.y * SIFT_EX_RCH;
    const int64_t lsz = (int64_t)w * h;
    const float* base = gauss + (int64_t)f * g.g_frame + g.g_ofs[o] + min(c, w - 2);
    const bool ok0 = lane >= 1 && c < w - SIFT_BORDER, ok1 = lane <= 62 && c + 1 < w - SIFT_BORDER;
    const float thr = (float)sp.threshold;
    float2 val[5][3], hx[5][3], hn[5][3];
    auto load_row = [&](int i, float2 (&v)[5]) {
        const int r = min(ry0 - 1 + i, h - 1);
        float2 gl[6];
#pragma unroll
        for (int L = 0; L < 6; ++L) __builtin_memcpy(&gl[L], base + lsz * L + (int64_t)r * w, 8);
#pragma unroll
        for (int L = 0; L < 5; ++L) v[L] = make_float2(gl[L + 1].x - gl[L].x, gl[L + 1].y - gl[L].y);
    };
    // candidates are staged per wave in LDS and appended to the frame's list with ONE returning global atomic when the wave is done:
    // a returning atomic inside the loop is followed by s_waitcnt vmcnt(0), i.e. by the whole latency of the 18 row loads requested
    // ahead — in the third of all row steps that hold a candidate
    __shared__ uint32_t s_stage[4][SIFT_EX_STAGE];
    uint32_t nst = 0;                                                              // (wave-uniform)
    auto emit = [&](bool ext, int L, int rc, int cc) {
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(ext);
        if (mk == 0ull) return;
        const uint32_t code = ((uint32_t)o << 28) | ((uint32_t)L << 26) | ((uint32_t)rc << 13) | (uint32_t)cc;
        const uint32_t pos = nst + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        if (ext) {
            if (pos < (uint32_t)SIFT_EX_STAGE) s_stage[wave][pos] = code;
            else {                                                                 // staging full (a chunk of noise): straight to the list
                const uint32_t slot = atomicAdd(&cand_count[f], 1u);
                if (slot >= (uint32_t)sp.cand_cap) atomicOr(flags, 16u);
                else cand[(size_t)f * sp.cand_cap + slot] = code;
            }
        }
        nst += (uint32_t)__builtin_popcountll(mk);
    };
    auto step = [&](auto slot_tag, int i, const float2 (&v)[5]) {
        constexpr int S = decltype(slot_tag)::value, S1 = (S + 2) % 3;           // S1 = the slot of the previous row (the centre row)
#pragma unroll
        for (int L = 0; L < 5; ++L) {
            const float ly = sift_wave_shr1(v[L].y), rx = sift_wave_shl1(v[L].x);  // left neighbour's right column, right neighbour's left column
            val[L][S] = v[L];
            hx[L][S] = make_float2(fmaxf(fmaxf(ly, v[L].x), v[L].y), fmaxf(fmaxf(v[L].x, v[L].y), rx));
            hn[L][S] = make_float2(fminf(fminf(ly, v[L].x), v[L].y), fminf(fminf(v[L].x, v[L].y), rx));
        }
        if (i < 2) return;
        const int rc = ry0 + i - 2;                                               // centre row of this step
        if (rc >= h - SIFT_BORDER) return;                                        // (uniform)
        float2 vx[5], vn[5];
#pragma unroll
        for (int L = 0; L < 5; ++L) {
            vx[L] = make_float2(fmaxf(fmaxf(hx[L][0].x, hx[L][1].x), hx[L][2].x), fmaxf(fmaxf(hx[L][0].y, hx[L][1].y), hx[L][2].y));
            vn[L] = make_float2(fminf(fminf(hn[L][0].x, hn[L][1].x), hn[L][2].x), fminf(fminf(hn[L][0].y, hn[L][1].y), hn[L][2].y));
        }
        // the six tests of this row (3 layers x 2 columns) branch-free — |x| > thr and (x > 0 ? x >= max27 : x <= min27) is
        // (x > thr and x >= max27) or (x < -thr and x <= min27) for thr >= 0 — and ONE branch for the rare row that holds an
        // extremum (a branch per test ran its body for almost every wave: some lane is above the threshold wherever there is an edge)
        bool e0[4], e1[4];
        bool any = false;
#pragma unroll
        for (int L = 1; L <= 3; ++L) {
            const float2 x = val[L][S1];
            const float mx0 = fmaxf(fmaxf(vx[L - 1].x, vx[L].x), vx[L + 1].x), mn0 = fminf(fminf(vn[L - 1].x, vn[L].x), vn[L + 1].x);
            const float mx1 = fmaxf(fmaxf(vx[L - 1].y, vx[L].y), vx[L + 1].y), mn1 = fminf(fminf(vn[L - 1].y, vn[L].y), vn[L + 1].y);
            e0[L] = ok0 & (((x.x > thr) & (x.x >= mx0)) | ((x.x < -thr) & (x.x <= mn0)));
            e1[L] = ok1 & (((x.y > thr) & (x.y >= mx1)) | ((x.y < -thr) & (x.y <= mn1)));
            any |= e0[L] | e1[L];
        }
        if (__builtin_amdgcn_ballot_w64(any) == 0ull) return;
#pragma unroll
        for (int L = 1; L <= 3; ++L) { emit(e0[L], L, rc, c); emit(e1[L], L, rc, c + 1); }
    };
    using S0 = std::integral_constant<int, 0>; using S1t = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;
    float2 va[5], vb[5], vc[5];
    load_row(0, va); load_row(1, vb); load_row(2, vc);
    for (int i = 0; i < SIFT_EX_RCH + 2; i += 3) {
        if (ry0 + i - 2 >= h - SIFT_BORDER) break;                                // (uniform: nothing left to emit)
        float2 na[5], nb[5], nc[5];                                               // the next three rows, requested before this trio is used
        load_row(i + 3, na); load_row(i + 4, nb); load_row(i + 5, nc);
        step(S0{}, i, va); step(S1t{}, i + 1, vb); step(S2{}, i + 2, vc);
#pragma unroll
        for (int L = 0; L < 5; ++L) { va[L] = na[L]; vb[L] = nb[L]; vc[L] = nc[L]; }
    }
    const uint32_t ns = min(nst, (uint32_t)SIFT_EX_STAGE);
    if (ns) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&cand_count[f], ns);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < ns; i += 64) {
            if (base + i >= (uint32_t)sp.cand_cap) { atomicOr(flags, 16u); break; }
            cand[(size_t)f * sp.cand_cap + base + i] = s_stage[wave][i];
        }
    }
}

// Matx33f::solve(b, DECOMP_LU) (matx.hpp Matx_FastSolveOp<float, 3, 3, 1>): Cramer's rule in f32; singular -> 0
__device__ __forceinline__ void sift_solve3(const float (&a)[3][3], const float (&b)[3], float (&x)[3]) {
    float d = a[0][0] * (a[1][1] * a[2][2] - a[2][1] * a[1][2]) - a[0][1] * (a[1][0] * a[2][2] - a[2][0] * a[1][2]) +
              a[0][2] * (a[1][0] * a[2][1] - a[2][0] * a[1][1]);
    x[0] = x[1] = x[2] = 0;
    if (d == 0) return;
    d = 1 / d;
    x[0] = d * (b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2]) + a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
    x[1] = d * (a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) - b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) + a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
    x[2] = d * (a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) - a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0]) + b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
}

__device__ __forceinline__ float sift_expf(float x) { return (float)exp((double)x); }

// grid (ceil(candidates / 256), n), block 256 = 4 waves, 64 candidates per wave
__global__ __launch_bounds__(256) void sift_refine_kernel(SiftGeom g, SiftParams sp, const float* __restrict__ gauss,
                                                          const uint32_t* __restrict__ cand, const uint32_t* __restrict__ cand_count,
                                                          SiftRaw* __restrict__ raw, uint32_t* __restrict__ raw_count, uint32_t* __restrict__ flags) {
    __shared__ unsigned long long s_hist[4][SIFT_BINS];
    __shared__ float s_sm[4][SIFT_BINS];
    const int f = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t ncand = min(cand_count[f], (uint32_t)sp.cand_cap);
    if ((blockIdx.x * 4 + wave) * 64u >= ncand) return;                 // (wave-uniform; no block barrier below)
    // Phase 1 — adjustLocalExtrema, ONE CANDIDATE PER LANE (a wave per candidate spent its time in the latency of 40 dependent
    // loads per Newton step on behalf of one lane; most candidates are rejected here).  Phase 2 — the wave walks over its
    // surviving lanes and builds each one's orientation histogram with all 64 lanes.
    const uint32_t ci = (blockIdx.x * 4 + wave) * 64u + lane;
    const bool have = ci < ncand;
    const uint32_t cw = cand[(size_t)f * sp.cand_cap + min(ci, ncand - 1)];
    int octv = (int)(cw >> 28);
    int layer = (int)((cw >> 26) & 3), r = (int)((cw >> 13) & 8191), c = (int)(cw & 8191);
    int w = g.ow[octv], h = g.oh[octv];
    int64_t lsz = (int64_t)w * h;
    const SiftDog dbase{gauss + (int64_t)f * g.g_frame + g.g_ofs[octv], lsz};
    const float img_scale = 1.f / 255.f, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale, cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int it = 0;
    bool ok = true;
    for (; it < SIFT_STEPS; ++it) {
        const SiftDog img = dbase + lsz * layer;
        const SiftDog prev = img - lsz;
        const SiftDog next = img + lsz;
        const int64_t p = (int64_t)r * w + c;
        const float dD[3] = {(img[p + 1] - img[p - 1]) * deriv_scale, (img[p + w] - img[p - w]) * deriv_scale, (next[p] - prev[p]) * deriv_scale};
        const float v2 = img[p] * 2;
        const float dxx = (img[p + 1] + img[p - 1] - v2) * second_deriv_scale;
        const float dyy = (img[p + w] + img[p - w] - v2) * second_deriv_scale;
        const float dss = (next[p] + prev[p] - v2) * second_deriv_scale;
        const float dxy = (img[p + w + 1] - img[p + w - 1] - img[p - w + 1] + img[p - w - 1]) * cross_deriv_scale;
        const float dxs = (next[p + 1] - next[p - 1] - prev[p + 1] + prev[p - 1]) * cross_deriv_scale;
        const float dys = (next[p + w] - next[p - w] - prev[p + w] + prev[p - w]) * cross_deriv_scale;
        const float H[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        float X[3];
        sift_solve3(H, dD, X);
        xi = -X[2]; xr = -X[1]; xc = -X[0];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        const float big = (float)(INT_MAX / 3);
        if (fabsf(xi) > big || fabsf(xr) > big || fabsf(xc) > big) { ok = false; break; }
        c += (int)rintf(xc); r += (int)rintf(xr); layer += (int)rintf(xi);
        if (layer < 1 || layer > SIFT_NL || c < SIFT_BORDER || c >= w - SIFT_BORDER || r < SIFT_BORDER || r >= h - SIFT_BORDER) { ok = false; break; }
    }
    bool alive = have && ok && it < SIFT_STEPS;
    slideo_keypoint kpt{};
    if (alive) {
        alive = false;
        do {
        const SiftDog img = dbase + lsz * layer;
        const SiftDog prev = img - lsz;
        const SiftDog next = img + lsz;
        const int64_t p = (int64_t)r * w + c;
        const float dD[3] = {(img[p + 1] - img[p - 1]) * deriv_scale, (img[p + w] - img[p - w]) * deriv_scale, (next[p] - prev[p]) * deriv_scale};
        const float t = dD[0] * xc + dD[1] * xr + dD[2] * xi;
        const float contr = img[p] * img_scale + t * 0.5f;
        if (fabsf(contr) * SIFT_NL < sp.contrast_threshold) break;
        const float v2 = img[p] * 2.f;
        const float dxx = (img[p + 1] + img[p - 1] - v2) * second_deriv_scale;
        const float dyy = (img[p + w] + img[p - w] - v2) * second_deriv_scale;
        const float dxy = (img[p + w + 1] - img[p + w - 1] - img[p - w + 1] + img[p - w - 1]) * cross_deriv_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        const float et = sp.edge_threshold;
        if (det <= 0 || tr * tr * et >= (et + 1) * (et + 1) * det) break;
        kpt.x = ((float)c + xc) * (float)(1 << octv);
        kpt.y = ((float)r + xr) * (float)(1 << octv);
        kpt.octave = octv + (layer << 8) + ((int)rint(((double)xi + 0.5) * 255) << 16);
        kpt.size = sp.sigma * (float)pow(2.0, (double)(((float)layer + xi) / SIFT_NL)) * (float)(1 << octv) * 2;      // (f64 pow rounded to f32, as the oracle)
        kpt.response = fabsf(contr);
        kpt.angle = 0;
        alive = true;
        } while (false);
    }
    // Phase 2: the survivors of this wave, one after the other (their state broadcast from the owning lane)
    unsigned long long todo = __builtin_amdgcn_ballot_w64(alive);
    const slideo_keypoint kpt_mine = kpt;
    const int r_mine = r, c_mine = c, layer_mine = layer, octv_mine = octv;
    while (todo) {
    const int src = __builtin_ctzll(todo);
    todo &= todo - 1;
    auto bi = [&](int v) -> int { return __builtin_amdgcn_readlane(v, src); };
    auto bf = [&](float v) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); };
    r = bi(r_mine); c = bi(c_mine); layer = bi(layer_mine); octv = bi(octv_mine);
    kpt.x = bf(kpt_mine.x); kpt.y = bf(kpt_mine.y); kpt.size = bf(kpt_mine.size); kpt.angle = 0; kpt.response = bf(kpt_mine.response); kpt.octave = bi(kpt_mine.octave);
    w = g.ow[octv]; h = g.oh[octv]; lsz = (int64_t)w * h;
    // orientation histogram on the Gaussian layer the extremum ended in
    const float scl_octv = kpt.size * 0.5f / (float)(1 << octv);
    const int radius = (int)rintf(4.5f * scl_octv);
    const float sigma = 1.5f * scl_octv;
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    const float* gimg = gauss + (int64_t)f * g.g_frame + g.g_ofs[octv] + lsz * layer;
    if (lane < SIFT_BINS) s_hist[wave][lane] = 0ull;
    __builtin_amdgcn_wave_barrier();
    const int side = 2 * radius + 1;
    for (int s = lane; s < side * side; s += 64) {
        const int i = s / side - radius, j = s - (s / side) * side - radius;
        const int y = r + i, x = c + j;
        if (y <= 0 || y >= h - 1 || x <= 0 || x >= w - 1) continue;
        const int64_t p = (int64_t)y * w + x;
        const float dx = gimg[p + 1] - gimg[p - 1], dy = gimg[p - w] - gimg[p + w];
        const float wgt = sift_expf((float)(i * i + j * j) * expf_scale);
        const float ori = fast_atan2f_cv(dy, dx, sp.atan_fma != 0), mag = sqrtf(dx * dx + dy * dy);
        int bin = (int)rintf((SIFT_BINS / 360.f) * ori);
        if (bin >= SIFT_BINS) bin -= SIFT_BINS;
        if (bin < 0) bin += SIFT_BINS;
        atomicAdd(&s_hist[wave][bin], (unsigned long long)(long long)llrint((double)(wgt * mag) * SIFT_FIXD));
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float hv = 0;
    if (lane < SIFT_BINS) {
        auto th = [&](int i) -> float { return (float)((double)(long long)s_hist[wave][(i + SIFT_BINS) % SIFT_BINS] / SIFT_FIXD); };
        hv = (th(lane - 2) + th(lane + 2)) * (1.f / 16.f) + (th(lane - 1) + th(lane + 1)) * (4.f / 16.f) + th(lane) * (6.f / 16.f);
        s_sm[wave][lane] = hv;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float omax = lane < SIFT_BINS ? hv : 0.f;               // (hist >= 0: the maximum over the 36 bins)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) omax = fmaxf(omax, __shfl_xor(omax, d));
    const float mag_thr = omax * 0.8f;
    if (lane < SIFT_BINS) {
        const int n = SIFT_BINS, j = lane;
        const int l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
        const float hl = s_sm[wave][l], hr = s_sm[wave][r2], hj = hv;
        if (hj > hl && hj > hr && hj >= mag_thr) {
            float bin = (float)j + 0.5f * (hl - hr) / (hl - 2 * hj + hr);
            bin = bin < 0 ? n + bin : bin >= n ? bin - n : bin;
            slideo_keypoint k2 = kpt;
            k2.angle = 360.f - (float)((360.f / n) * bin);
            if (fabsf(k2.angle - 360.f) < FLT_EPSILON) k2.angle = 0.f;
            const uint32_t slot = atomicAdd(&raw_count[f], 1u);
            if (slot < (uint32_t)sp.raw_cap) {
                SiftRaw o;
                o.kp = k2;
                o.key = ((uint64_t)octv << 34) | ((uint64_t)layer << 32) | ((uint64_t)r << 19) | ((uint64_t)c << 6) | (uint64_t)j;
                raw[(size_t)f * sp.raw_cap + slot] = o;
            } else atomicOr(flags, 32u);
        }
    }
    __builtin_amdgcn_wave_barrier();                                    // (s_hist / s_sm are reused by the next survivor)
    }   // survivors
}

// One block of 1024 per frame.  items: [n][raw_cap] u64 workspace; kept: [n][raw_cap] u32 (slots of the kept keypoints in
// canonical order); kept_count[n].
__global__ __launch_bounds__(1024) void sift_select_kernel(SiftParams sp, const SiftRaw* __restrict__ raw, const uint32_t* __restrict__ raw_count,
                                                           uint64_t* __restrict__ items, uint32_t* __restrict__ kept, uint32_t* __restrict__ kept_count) {
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_prefix, s_need, s_n;
    __shared__ uint32_t s_scan[1024];
    const int f = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = min(raw_count[f], (uint32_t)sp.raw_cap);
    const SiftRaw* R = raw + (size_t)f * sp.raw_cap;
    uint64_t* a = items + (size_t)f * sp.raw_cap;
    for (uint32_t i = tid; i < n; i += 1024) a[i] = (R[i].key << 16) | (uint64_t)i;          // (raw_cap <= 65536)
    __syncthreads();
    // canonical order: the bitonic network of sort_global_kernel on the frame's items (virtual +infinity tail)
    uint32_t np = 2;
    while (np < n) np <<= 1;
    auto cmpx = [&](uint32_t lo, uint32_t hi) {
        if (hi < n) {
            const uint64_t x = a[lo], y = a[hi];
            if (x > y) { a[lo] = y; a[hi] = x; }
        }
    };
    if (n > 1)
        for (uint32_t k = 2; k <= np; k <<= 1) {
            const uint32_t hk = k >> 1;
            for (uint32_t i = tid; i < np / 2; i += 1024) { const uint32_t base = (i / hk) * k, t = i % hk; cmpx(base + t, base + k - 1 - t); }
            __syncthreads();
            for (uint32_t j = hk >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < np / 2; i += 1024) { const uint32_t lo = (i / j) * 2 * j + (i % j); cmpx(lo, lo + j); }
                __syncthreads();
            }
        }
    // unique keys (equal keys = two extrema refined to the same keypoint: exact duplicates); uniq(i) = first of its run
    auto uniq = [&](uint32_t i) -> bool { return i == 0 || (a[i] >> 16) != (a[i - 1] >> 16); };
    auto resp_bits = [&](uint32_t i) -> uint32_t { return __float_as_uint(R[(uint32_t)(a[i] & 0xFFFFu)].kp.response); };   // response >= 0: bit order = value order
    // number of unique keypoints
    uint32_t cnt = 0;
    for (uint32_t i = tid; i < n; i += 1024) cnt += uniq(i) ? 1u : 0u;
    s_scan[tid] = cnt;
    __syncthreads();
    if (tid == 0) { uint32_t t = 0; for (int i = 0; i < 1024; ++i) t += s_scan[i]; s_n = t; s_prefix = 0; s_need = (uint32_t)max(sp.nfeatures, 0); }
    __syncthreads();
    uint32_t thr_bits = 0;                                 // keep response >= thr
    if (sp.nfeatures > 0 && s_n > (uint32_t)sp.nfeatures) {
        // radix select of the nfeatures-th largest response: 4 passes of 8 bits from the top
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix;
            for (uint32_t i = tid; i < n; i += 1024)
                if (uniq(i)) {
                    const uint32_t b = resp_bits(i);
                    if (pass == 0 || (b >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&s_hist[(b >> shift) & 255u], 1u);
                }
            __syncthreads();
            if (tid == 0) {
                uint32_t need = s_need, d = 255;
                for (;; --d) { if (s_hist[d] >= need) break; need -= s_hist[d]; if (d == 0) break; }
                s_need = need; s_prefix = prefix | (d << shift);
            }
            __syncthreads();
        }
        thr_bits = s_prefix;
    }
    // compaction in canonical order
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t b0 = min(n, tid * per), b1 = min(n, b0 + per);
    cnt = 0;
    for (uint32_t i = b0; i < b1; ++i) cnt += (uniq(i) && resp_bits(i) >= thr_bits) ? 1u : 0u;
    s_scan[tid] = cnt;
    __syncthreads();
    if (tid == 0) { uint32_t run = 0; for (int i = 0; i < 1024; ++i) { const uint32_t v = s_scan[i]; s_scan[i] = run; run += v; } kept_count[f] = run; }
    __syncthreads();
    uint32_t o = s_scan[tid];
    uint32_t* K = kept + (size_t)f * sp.raw_cap;
    for (uint32_t i = b0; i < b1; ++i)
        if (uniq(i) && resp_bits(i) >= thr_bits) K[o++] = (uint32_t)(a[i] & 0xFFFFu);
}

// grid (ceil(total_cap / 4)), block 256 = 4 waves, one kept keypoint per wave; qofs from scan_kernel over kept_count.
// Writes the keypoint (scaled back to the input image) and its 128 descriptor bytes.
__global__ __launch_bounds__(256) void sift_describe_kernel(SiftGeom g, SiftParams sp, int nframes, const float* __restrict__ gauss,
                                                            const SiftRaw* __restrict__ raw, const uint32_t* __restrict__ kept,
                                                            const uint32_t* __restrict__ qofs, slideo_keypoint* __restrict__ kp_out,
                                                            uint8_t* __restrict__ desc_out) {
    __shared__ unsigned long long s_h[4][360];
    __shared__ float s_v[4][128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t gi = blockIdx.x * 4 + wave;
    const uint32_t total = qofs[nframes];
    if (gi >= total) return;
    int f = 0;
    { int lo = 0, hi = nframes; while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (qofs[mid] <= gi) lo = mid; else hi = mid; } f = lo; }
    const uint32_t slot = kept[(size_t)f * sp.raw_cap + (gi - qofs[f])];
    slideo_keypoint k = raw[(size_t)f * sp.raw_cap + slot].kp;
    const int octave = k.octave & 255, layer = (k.octave >> 8) & 255;
    const float scale = 1.f / (float)(1 << octave);
    const float size = k.size * scale;
    float ori = 360.f - k.angle;
    if (fabsf(ori - 360.f) < FLT_EPSILON) ori = 0.f;
    const float ptx = k.x * scale, pty = k.y * scale, scl = size * 0.5f;
    const int w = g.ow[octave], h = g.oh[octave];
    const float* img = gauss + (int64_t)f * g.g_frame + g.g_ofs[octave] + (int64_t)w * h * layer;
    constexpr int d = 4, n = 8;
    const int px = (int)rintf(ptx), py = (int)rintf(pty);
    float cos_t = (float)cos((double)(ori * (float)(3.14159265358979323846 / 180)));
    float sin_t = (float)sin((double)(ori * (float)(3.14159265358979323846 / 180)));
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = 3.f * scl;
    int radius = (int)rintf(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    radius = min(radius, (int)sqrt((double)w * w + (double)h * h));
    cos_t /= hist_width; sin_t /= hist_width;
    for (int i = lane; i < 360; i += 64) s_h[wave][i] = 0ull;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int side = 2 * radius + 1;
    for (int s = lane; s < side * side; s += 64) {
        const int i = s / side - radius, j = s - (s / side) * side - radius;
        const float c_rot = (float)j * cos_t - (float)i * sin_t, r_rot = (float)j * sin_t + (float)i * cos_t;
        float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
        const int r = py + i, c = px + j;
        if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < h - 1 && c > 0 && c < w - 1)) continue;
        const int64_t p = (int64_t)r * w + c;
        const float dx = img[p + 1] - img[p - 1], dy = img[p - w] - img[p + w];
        const float wg = sift_expf((c_rot * c_rot + r_rot * r_rot) * exp_scale);
        const float o = fast_atan2f_cv(dy, dx, sp.atan_fma != 0);
        float obin = (o - ori) * bins_per_rad;
        const float mag = sqrtf(dx * dx + dy * dy) * wg;
        const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
        int o0 = (int)floorf(obin);
        rbin -= (float)r0; cbin -= (float)c0; obin -= (float)o0;
        if (o0 < 0) o0 += n;
        if (o0 >= n) o0 -= n;
        const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
        const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
        const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111, v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
        const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011, v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
        const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
        auto add = [&](int kx, float v) { atomicAdd(&s_h[wave][kx], (unsigned long long)(long long)llrint((double)v * SIFT_FIXD)); };
        add(idx, v_rco000); add(idx + 1, v_rco001); add(idx + (n + 2), v_rco010); add(idx + (n + 3), v_rco011);
        add(idx + (d + 2) * (n + 2), v_rco100); add(idx + (d + 2) * (n + 2) + 1, v_rco101);
        add(idx + (d + 3) * (n + 2), v_rco110); add(idx + (d + 3) * (n + 2) + 1, v_rco111);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int e = lane; e < 128; e += 64) {
        const int cell = e >> 3, kk = e & 7, i = cell >> 2, j = cell & 3;
        const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
        long long hq = (long long)s_h[wave][idx + kk];
        if (kk < 2) hq += (long long)s_h[wave][idx + n + kk];           // circular orientation bins
        s_v[wave][e] = (float)((double)hq / SIFT_FIXD);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the two norms in the oracle's serial order (every lane the same 256 adds)
    float nrm2 = 0;
    for (int e = 0; e < 128; ++e) { const float v = s_v[wave][e]; nrm2 += v * v; }
    const float thr = sqrtf(nrm2) * 0.2f;
    nrm2 = 0;
    for (int e = 0; e < 128; ++e) { const float v = fminf(s_v[wave][e], thr); nrm2 += v * v; }
    nrm2 = 512.f / fmaxf(sqrtf(nrm2), FLT_EPSILON);
    for (int e = lane; e < 128; e += 64) {
        const int q = (int)rintf(fminf(s_v[wave][e], thr) * nrm2);
        desc_out[(size_t)gi * 128 + e] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
    if (lane == 0) {
        k.octave = (k.octave & ~255) | ((k.octave - 1) & 255);
        k.x *= 0.5f; k.y *= 0.5f; k.size *= 0.5f;
        kp_out[gi] = k;
    }
}

}  // namespace slideo
