// orb.hip.h — ORB detect + describe kernels for gfx950.
//
// Replaces FeatureExtractor::find_keypoints_and_descriptors
// (crates/matching-opencv/src/feature_extractor.rs:29-46), i.e. OpenCV 4.5.2
// ORB_Impl::detectAndCompute as restated in SURVEY.md Appendix A.1-A.7.
// All stages are batched over the frames of a batch (blockIdx.y / .z = frame)
// so that one launch covers >> 256 workgroups.
//
//   gray_kernel        BGR8 -> level 0                      (HBM: 3wh in, wh out)
//   resize_quad_kernel level l-1 -> l, INTER_LINEAR_EXACT   (7 dependent launches; per-group tables, shrink factors < ~2.3;
//                      resize_kernel is the same arithmetic for any factor)
//   fast_kernel        FAST-9/16 score + 3x3 NMS + border filter -> candidate
//                      list + per-(frame,level) score histogram   (all levels, one launch)
//   blur_kernel        7x7 sigma-2 fixed-point Gaussian     (all levels, one launch)
//   threshold_kernel   retainBest: per-level score threshold from the histogram
//                      (ties kept => order independent, deterministic)
//   scan_kernel        per-frame keypoint offsets
//   compact_kernel     candidates >= threshold -> per-frame keypoint items
//   sort_kernel        canonical order (octave, y, x) — bitonic sort in LDS
//   describe_kernel    IC angle (unblurred level) + rotated BRIEF-256 (blurred level)
//
// Keypoints sit >= edge_threshold (62) px from the level edge, the IC disc has
// radius 31 and the rotated BRIEF samples radius <= 44 (+3 blur taps), so
// neither stage ever reads outside the level: no reflected border is
// materialised (OpenCV stores one of 63 px; it is never sampled).
#pragma once
#include <limits.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "geom.h"
#include "types.h"
#include "cv_math.hip.h"

namespace slideo {

// compile-time experiment hook (-DORB_PRIO=n, tools/ab_matrix.sh; not in the product build): the ORB stage's waves at instruction priority n
// (the search's matrix sections run at 1 / 2, everything else at 0)
#ifdef ORB_PRIO
#define SLIDEO_ORB_PRIO() __builtin_amdgcn_s_setprio(ORB_PRIO)
#else
#define SLIDEO_ORB_PRIO()
#endif

// ---------------------------------------------------------------------------
// [OCV A.1] gray = (B*3735 + G*19235 + R*9798 + 2^14) >> 15
// grid (ceil(w/4/256), h, B)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gray_kernel(const uint8_t* __restrict__ frames, int64_t frame_stride,
                                                   int stride, uint8_t* __restrict__ pyr, int64_t pyr_frame_bytes,
                                                   int w, int h, int pitch, int aligned4, GrayCoef gc) {
    SLIDEO_ORB_PRIO();
    const uint32_t half = 1u << (gc.shift - 1);
    // (coefficients < 2^15, pixels < 2^8: 24-bit multiplies — full rate, where v_mul_lo_u32 takes four issue slots)
    auto gray_of = [&](uint32_t b, uint32_t g, uint32_t r) { return (__umul24(b, gc.cb) + __umul24(g, gc.cg) + __umul24(r, gc.cr) + half) >> gc.shift; };
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* src = frames + (int64_t)blockIdx.z * frame_stride + (int64_t)y * stride + (int64_t)x * 3;
    uint8_t* dst = pyr + (int64_t)blockIdx.z * pyr_frame_bytes + (int64_t)y * pitch + x;
    if (aligned4 && x + 3 < w) {
        const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
        uint32_t a = s[0], b = s[1], c = s[2];   // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
        uint32_t g0 = gray_of(a & 255, (a >> 8) & 255, (a >> 16) & 255);
        uint32_t g1 = gray_of(a >> 24, b & 255, (b >> 8) & 255);
        uint32_t g2 = gray_of((b >> 16) & 255, b >> 24, c & 255);
        uint32_t g3 = gray_of((c >> 8) & 255, (c >> 16) & 255, c >> 24);
        *reinterpret_cast<uint32_t*>(dst) = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    } else {
        for (int i = 0; i < 4 && x + i < w; ++i)
            dst[i] = (uint8_t)gray_of(src[3 * i], src[3 * i + 1], src[3 * i + 2]);
    }
}

// ---------------------------------------------------------------------------
// [OCV A.2] INTER_LINEAR_EXACT: out = (cy0*(cx0*p00+cx1*p01) + cy1*(cx0*p10+cx1*p11) + 2^15) >> 16
// ---------------------------------------------------------------------------
// One thread makes 4 adjacent outputs of one row; threads are numbered flat over (row, 4-pixel group)
// so every wave is full whatever the level width.  The 4 outputs read <= 12 source bytes per row
// (shrink factors < 2), fetched as 3 aligned dwords; the two taps of an output are picked with one
// v_perm_b32 (selector built from the byte offset) and weighted with one v_dot2_u32_u16 against the
// pre-packed (256-c1, c1) pair.  grid (ceil(nxq*dh/256), 1, B), nxq = ceil(dw/4).
__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t b, uint32_t c = 0u) {      // a.lo*b.lo + a.hi*b.hi + c
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
}

// A thread makes the same 4 columns of RESIZE_ROWS consecutive output rows: the x tables are fetched once, and
// all of the thread's row-table entries, then all of its 6 x RESIZE_ROWS source dwords, are in flight together
// (the kernel is latency bound: table -> address -> data is two dependent round trips per output dword).
constexpr int RESIZE_ROWS = 4;

__global__ __launch_bounds__(256) void resize_kernel(uint8_t* __restrict__ pyr, int64_t pyr_frame_bytes,
                                                     LevelGeom src, LevelGeom dst,
                                                     const uint32_t* __restrict__ lin_tab, int nxq, uint32_t nxq_magic) {
    SLIDEO_ORB_PRIO();
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const int yg = nxq_magic ? (int)__umulhi(t, nxq_magic) : (int)t;     // t / nxq (exact, see build_pyr_geom)
    const int y0 = yg * RESIZE_ROWS;
    if (y0 >= dst.h) return;
    const int x0 = ((int)t - (int)__umul24(yg, nxq)) * 4;
    uint8_t* base = pyr + (int64_t)blockIdx.z * pyr_frame_bytes;
    const int nx = min(4, dst.w - x0);
    // both x tables are 16-byte aligned and padded to a multiple of 4 entries with copies of the last one
    const uint4 xq = *reinterpret_cast<const uint4*>(lin_tab + dst.xtab_ofs + x0);
    const uint4 cq = *reinterpret_cast<const uint4*>(lin_tab + dst.xctab_ofs + x0);
    uint32_t ye[RESIZE_ROWS];
#pragma unroll
    for (int r = 0; r < RESIZE_ROWS; ++r) ye[r] = lin_tab[dst.ytab_ofs + min(y0 + r, dst.h - 1)];
    const uint32_t xe[4] = {xq.x, xq.y, xq.z, xq.w};
    const uint32_t cp[4] = {cq.x, cq.y, cq.z, cq.w};
    // The 4 outputs tap at most 8 consecutive source bytes per row (shrink factors < 2.3): ONE unaligned 8-byte load per source
    // row, starting at the first output's left tap (clamped so that it stays inside the row's pitch), instead of three aligned
    // dwords and a per-tap choice between them — a third of the load instructions and half the VALU of the tap selection.
    const int xs = min((int)(xe[0] & 0xffff), src.pitch - 8);
    const bool narrow = (int)(xe[3] & 0xffff) + 1 - xs < 8 || ((int)(xe[3] & 0xffff) - xs < 8 && (xe[3] >> 16) == 0);
    uint2 a[RESIZE_ROWS], b[RESIZE_ROWS];              // (loaded unconditionally: a conditionally filled array would live in scratch)
#pragma unroll
    for (int r = 0; r < RESIZE_ROWS; ++r) {
        const int yo = ye[r] & 0xffff;
        const uint8_t* p0 = base + src.ofs + __umul24(yo, src.pitch) + xs;
        const uint8_t* p1 = base + src.ofs + __umul24(min(yo + 1, src.h - 1), src.pitch) + xs;
        __builtin_memcpy(&a[r], p0, 8);
        __builtin_memcpy(&b[r], p1, 8);
    }
#pragma unroll
    for (int r = 0; r < RESIZE_ROWS; ++r) {
        const int y = y0 + r;
        if (y >= dst.h) break;
        const int yo = ye[r] & 0xffff, cy1 = ye[r] >> 16, cy0 = 256 - cy1;
        uint32_t v[4];
        if (narrow) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // tap 2 is byte p + 1 even at the right edge: there c1 == 0, so its value is irrelevant (p + 1 = 8 selects a
                // constant byte)
                const uint32_t pq = (uint32_t)((int)(xe[i] & 0xffff) - xs);                  // 0..7
                const uint32_t sel = pq * 0x00010001u + 0x0c010c00u;                         // bytes (p, 0, p + 1, 0)
                const uint32_t ta = __builtin_amdgcn_perm(a[r].y, a[r].x, sel);
                const uint32_t tb = __builtin_amdgcn_perm(b[r].y, b[r].x, sel);
                v[i] = __umul24(cy0, dot2_u16(ta, cp[i])) + (__umul24(cy1, dot2_u16(tb, cp[i])) + (1u << 15));   // 9 x 16 bits: full-rate 24-bit multiplies
            }
        } else {      // (shrink factors >= 2.3; kept for generality)
            const uint8_t* r0 = base + src.ofs + __umul24(yo, src.pitch);
            const uint8_t* r1 = base + src.ofs + __umul24(min(yo + 1, src.h - 1), src.pitch);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int xo = xe[i] & 0xffff, cx1 = xe[i] >> 16, cx0 = 256 - cx1;
                const int xo1 = min(xo + 1, src.w - 1);
                uint32_t h0 = (uint32_t)cx0 * r0[xo] + (uint32_t)cx1 * r0[xo1];
                uint32_t h1 = (uint32_t)cx0 * r1[xo] + (uint32_t)cx1 * r1[xo1];
                v[i] = (uint32_t)cy0 * h0 + (uint32_t)cy1 * h1 + (1u << 15);
            }
        }
        // result byte = bits 16..23 of each v (v < 2^24)
        const uint32_t outv = __builtin_amdgcn_perm(__builtin_amdgcn_perm(v[3], v[2], 0x0c0c0602u),
                                                    __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u), 0x05040100u);
        uint8_t* d = base + dst.ofs + __umul24(y, dst.pitch) + x0;
        if (nx == 4) *reinterpret_cast<uint32_t*>(d) = outv;
        else for (int i = 0; i < nx; ++i) d[i] = (uint8_t)(outv >> (8 * i));
    }
}

// The same arithmetic for levels whose every group of 4 outputs taps inside 8 consecutive source bytes (LevelGeom::rq_ok: all
// shrink factors below ~2.3, i.e. every ORB pyramid).  resize_kernel spends about 60 VALU instructions per 4 outputs, a third
// of them on what does not depend on the row: the window start, the tap selectors, the choice between its two tap paths, 64-bit
// addresses.  Here those come from per-group tables (geom.h: xs, selectors; the packed weights are resize_kernel's), the four
// rows' y entries are one 16-byte load, addresses are 32-bit offsets from a scalar base, and nothing branches but the store of
// a row past the level's last one: per output row 8 v_perm, 8 v_dot2, 8 multiplies, 4 adds, 3 perms to pack.
// grid (ceil(nxq * ceil(dh / 4) / 256), 1, B), the flat thread numbering of resize_kernel.
__global__ __launch_bounds__(256) void resize_quad_kernel(uint8_t* __restrict__ pyr, int64_t pyr_frame_bytes,
                                                          LevelGeom src, LevelGeom dst,
                                                          const uint32_t* __restrict__ lin_tab, int nxq, uint32_t nxq_magic) {
    SLIDEO_ORB_PRIO();
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t yg = nxq_magic ? __umulhi(t, nxq_magic) : t;
    const uint32_t y0 = yg * 4u;
    if ((int)y0 >= dst.h) return;
    const uint32_t q = t - __umul24(yg, (uint32_t)nxq);
    const uint8_t* __restrict__ sbase = pyr + (int64_t)blockIdx.z * pyr_frame_bytes + src.ofs;      // (scalar)
    uint8_t* __restrict__ dbase = pyr + (int64_t)blockIdx.z * pyr_frame_bytes + dst.ofs;
    const uint4 sel4 = reinterpret_cast<const uint4*>(lin_tab + dst.xsel_ofs)[q];
    const uint4 cp4 = reinterpret_cast<const uint4*>(lin_tab + dst.xctab_ofs)[q];
    const uint32_t xs = lin_tab[dst.xs_ofs + q];
    const uint4 ye4 = *reinterpret_cast<const uint4*>(lin_tab + dst.ytab_ofs + y0);     // (padded with copies of the last row's entry)
    const uint32_t ye[4] = {ye4.x, ye4.y, ye4.z, ye4.w};
    const uint32_t sel[4] = {sel4.x, sel4.y, sel4.z, sel4.w}, cp[4] = {cp4.x, cp4.y, cp4.z, cp4.w};
    uint2 a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t yo = ye[r] & 0xffffu;
        const uint32_t o0 = __umul24(yo, (uint32_t)src.pitch) + xs;
        const uint32_t o1 = __umul24(min(yo + 1u, (uint32_t)(src.h - 1)), (uint32_t)src.pitch) + xs;
        // Three ALIGNED dwords per source row and a funnel shift, not one unaligned 8-byte load: the texture path splits every
        // unaligned dwordx2 of a lane, and it — not HBM, not the VALU — was what bound the kernel (level 1 of 256 1080p frames:
        // 337 -> 232 us).  The twelve bytes may end 4 bytes past the row's pitch (the next row, the next level, or the slack
        // after the last frame's pyramid: orb_stage1 reserves it); the 8 bytes from xs on that are used lie inside the pitch.
        {
            const uint32_t sh = (xs & 3u) * 8u;
            uint32_t w0[3], w1[3];
            __builtin_memcpy(w0, sbase + (o0 & ~3u), 12);
            __builtin_memcpy(w1, sbase + (o1 & ~3u), 12);
            a[r] = make_uint2(__builtin_amdgcn_alignbit(w0[1], w0[0], sh), __builtin_amdgcn_alignbit(w0[2], w0[1], sh));
            b[r] = make_uint2(__builtin_amdgcn_alignbit(w1[1], w1[0], sh), __builtin_amdgcn_alignbit(w1[2], w1[1], sh));
        }
    }
    const uint32_t dofs = __umul24(y0, (uint32_t)dst.pitch) + 4u * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t cy1 = ye[r] >> 16, cy0 = 256u - cy1;
        uint32_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t ta = __builtin_amdgcn_perm(a[r].y, a[r].x, sel[i]);
            const uint32_t tb = __builtin_amdgcn_perm(b[r].y, b[r].x, sel[i]);
            v[i] = __umul24(cy0, dot2_u16(ta, cp[i])) + (__umul24(cy1, dot2_u16(tb, cp[i])) + (1u << 15));
        }
        const uint32_t outv = __builtin_amdgcn_perm(__builtin_amdgcn_perm(v[3], v[2], 0x0c0c0602u),
                                                    __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u), 0x05040100u);
        // (a whole dword also for the level's last, partial group: the pitch is a multiple of 16, the bytes past column w - 1 are
        // padding no kernel taps — and they come out the same every time: the tables repeat their last entry)
        if ((int)(y0 + r) < dst.h) *reinterpret_cast<uint32_t*>(dbase + (dofs + (uint32_t)r * (uint32_t)dst.pitch)) = outv;
    }
}

// ---------------------------------------------------------------------------
// [OCV A.3] FAST-9/16 score + NMS + runByImageBorder, all levels in one launch.
// grid (fast_tiles, B), block 256.  Candidate entry = score << 24 | y << 12 | x.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool run9(uint32_t m) {   // 9 contiguous set bits, circular over 16
    m |= m << 16;
    m &= m >> 1; m &= m >> 2; m &= m >> 4; m &= m >> 1;
    return m != 0;
}

__device__ __forceinline__ int level_of_tile(const PyrGeom& g, int tile, bool blur) {
    int l = 0;
    for (int i = 1; i < g.nlevels; ++i) {
        int t0 = blur ? g.lv[i].btile0 : g.lv[i].ftile0;
        int n = blur ? g.lv[i].btx * g.lv[i].bty : g.lv[i].ftx * g.lv[i].fty;
        if (n > 0 && tile >= t0) l = i;
    }
    return l;
}

// Tile = FAST_TW x FAST_TH outputs; scores are needed on a 1-px halo (FAST_SW x FAST_SH, FAST_SW = 128 so that a
// score row is exactly two wave-widths) and raw pixels on a 4-px halo.  The raw tile starts at pixel x0 - 5 (unaligned
// dword loads from the pyramid), so that score position sx sits at raw column sx + FAST_XO with FAST_XO = 4: the four
// positions 4j .. 4j + 3 of a row, and the pixels three rows above and below them, are each ONE aligned LDS dword.
constexpr int FAST_SW = FAST_TW + 2, FAST_SH = FAST_TH + 2;   // score tile (halo 1)
constexpr int FAST_XO = 4;                                    // raw column of score position 0
// raw tile: columns 0 .. FAST_SW - 1 + FAST_XO + 3 are needed (135); the row pitch is 35 dwords — ODD, so that the gathers of
// phases A2 (dwords around a group) and B (bytes around a position), whose lanes often sit in one column on consecutive rows (a
// vertical edge), spread over the banks (34 dwords: LDS bank-conflict cycles x 5)
constexpr int FAST_RW = FAST_TW + 14, FAST_RH = FAST_TH + 8;
static_assert(FAST_SW == 128 && FAST_RW % 4 == 0 && (FAST_RW / 4) % 2 == 1 && FAST_RW >= FAST_SW + FAST_XO + 3, "fast_kernel maps one score row onto two wave-widths");

typedef short fast_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short fast_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fast_s2 fast_swap(fast_s2 a) { return __builtin_shufflevector(a, a, 1, 0); }

// A block walks FAST_TPB consecutive tiles.  The raw pixels of tile i+1 are fetched into registers right after
// tile i's have been committed to LDS, so the global-load latency (the longest single wait of a tile: about 10 k of
// its 25 k cycles, per-wave s_memtime profile) overlaps tile i's three phases.
#ifndef FAST_TPB_V
#define FAST_TPB_V 8
#endif
constexpr int FAST_TPB = FAST_TPB_V;
constexpr int FAST_STAGE_CAP = 128;                                    // survivors staged per block before one list append (a tile keeps ~10; LDS: 8 blocks per CU)
constexpr int FAST_NDW = FAST_RW / 4;                                  // dwords per raw row
constexpr int FAST_NLD = (FAST_RH + 6) / 7;                            // raw dwords per loading thread and tile (fast_issue_loads)

struct FastTile { int l, x0, y0; };

// host-side twin of fast_tile_geo: the per-tile table fast_kernel reads (one 16-byte scalar load per tile instead of a loop
// over the levels' descriptors and an integer division)
inline int4 fast_tile_entry(const PyrGeom& g, int tile_id) {
    int l = 0;
    for (int i = 1; i < g.nlevels; ++i) if (g.lv[i].ftx * g.lv[i].fty > 0 && tile_id >= g.lv[i].ftile0) l = i;
    const LevelGeom& L = g.lv[l];
    const int tile = tile_id - L.ftile0, ty = tile / (L.ftx > 0 ? L.ftx : 1), tx = tile - ty * L.ftx;
    const int x0 = L.rx0 + tx * FAST_TW, y0 = L.ry0 + ty * FAST_TH;
    return make_int4(l, x0, y0, 0);
}

__device__ __forceinline__ FastTile fast_tile_geo(const int4* __restrict__ tab, int tile_id) {
    const int4 e = tab[tile_id];
    FastTile T;
    T.l = e.x; T.x0 = e.y; T.y0 = e.z;
    return T;
}

__device__ __forceinline__ FastTile fast_tile_geo(const PyrGeom& g, int tile_id) {
    FastTile T;
    T.l = level_of_tile(g, tile_id, false);
    const LevelGeom& L = g.lv[T.l];
    const int tile = tile_id - L.ftile0;
    const int ty = tile / L.ftx, tx = tile - ty * L.ftx;
    T.x0 = L.rx0 + tx * FAST_TW; T.y0 = L.ry0 + ty * FAST_TH;
    return T;
}

// raw tile: rows y0-4 .. y0+TH+3, columns from x0 - 1 - FAST_XO: UNALIGNED dword loads (x0 >= edge_threshold >= 5, so the first
// byte is inside the row), the byte offset clamped to the row's pitch — a clamped dword holds shifted pixels, all of them right
// of the level's last column but 3, which no needed position taps (those lie edge_threshold inside).  Rows clamped to the level.
// Who loads what: thread t < FAST_LT = 7 * FAST_NDW owns dword column t % FAST_NDW of raw rows t / FAST_NDW + 7 k, k = 0 .. FAST_NLD - 1
// (rows past FAST_RH - 1 repeat the last one and are dropped; the last few threads of the block idle here).  The k-th dword then
// sits 7 k rows below the first — one register (`rel0`, the first dword's offset inside the tile; it changes with the level's pitch
// only) and a scalar stride instead of an offset per dword — and lands in LDS at dword t + FAST_LT k.  A tile whose raw
// rectangle lies inside the level's rows and pitch — every tile but those of a level's last tile column — costs ONE addition
// per load; the others clamp as described above.
constexpr int FAST_LT = 7 * FAST_NDW;
static_assert(FAST_LT <= 256 && 7 * FAST_NLD >= FAST_RH, "fast_kernel's loader covers the raw tile");

__device__ __forceinline__ void fast_issue_loads(const PyrGeom& g, const FastTile& T, const uint8_t* __restrict__ frame_pyr,
                                                 uint32_t (&v)[FAST_NLD], uint32_t& rel0, int& rel_level) {
    const LevelGeom& L = g.lv[T.l];
    const uint8_t* img = frame_pyr + L.ofs;
    const int xs = T.x0 - 1 - FAST_XO, maxo = L.pitch - 4;
    int tid = min((int)threadIdx.x, FAST_LT - 1);
    if (T.l != rel_level) {                                                  // (wave-uniform)
        asm volatile("" : "+v"(tid));            // (opaque: the row / column split is redone here, not kept in registers for the whole kernel)
        const int rg = tid / FAST_NDW, c = tid - rg * FAST_NDW;
        rel0 = __umul24((uint32_t)rg, (uint32_t)L.pitch) + 4u * (uint32_t)c;
        rel_level = T.l;
    }
    if (T.y0 >= 4 && T.y0 - 4 + 7 * FAST_NLD <= L.h && xs + FAST_RW <= L.pitch) {      // (wave-uniform; rows up to 7 FAST_NLD - 1 are read)
        const uint32_t base = __umul24((uint32_t)(T.y0 - 4), (uint32_t)L.pitch) + (uint32_t)xs, step = 7u * (uint32_t)L.pitch;
#pragma unroll
        for (int k = 0; k < FAST_NLD; ++k) __builtin_memcpy(&v[k], img + (rel0 + (base + step * (uint32_t)k)), 4);
        return;
    }
    asm volatile("" : "+v"(tid));
    const int rg = tid / FAST_NDW, c = tid - rg * FAST_NDW;
#pragma unroll
    for (int k = 0; k < FAST_NLD; ++k) {
        const int gy = min(max(T.y0 - 4 + min(rg + 7 * k, FAST_RH - 1), 0), L.h - 1);
        __builtin_memcpy(&v[k], img + (__umul24((uint32_t)gy, (uint32_t)L.pitch) + (uint32_t)min(xs + 4 * c, maxo)), 4);   // (32-bit offset from the level's scalar base)
    }
}

#ifdef FAST_WAVES_EU      /* compile-time experiment hook: a register cap (6 waves per SIMD = 80 registers spills the tile offsets) */
#define FAST_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(FAST_WAVES_EU)))
#elif defined(FAST_VGPRS)  /* experiment hook: a HARD cap (amdgpu_num_vgpr counts in units of two: 32 = 64 registers; what does not fit spills) */
#define FAST_KERNEL_ATTR __attribute__((amdgpu_num_vgpr(FAST_VGPRS)))
#else
#define FAST_KERNEL_ATTR
#endif
__global__ FAST_KERNEL_ATTR __launch_bounds__(256) void fast_kernel(PyrGeom g, const uint8_t* __restrict__ pyr,
                                                   uint32_t* __restrict__ cand, uint32_t* __restrict__ cand_count,
                                                   uint32_t* __restrict__ hist, const int4* __restrict__ tile_tab) {
    SLIDEO_ORB_PRIO();
    __shared__ __attribute__((aligned(16))) uint8_t raw[FAST_RH][FAST_RW];
    __shared__ __attribute__((aligned(16))) uint8_t sc[FAST_SH][FAST_SW];
    // a wave's private share of the queue: its score rows (sy % 4 == wave), back to back
    constexpr int FAST_QR0 = (FAST_SH + 3) / 4, FAST_QR1 = (FAST_SH + 2) / 4, FAST_QR2 = (FAST_SH + 1) / 4;
    constexpr int FAST_PQ_TOTAL = 4 * 64 * (((FAST_SH * 32 + 255) / 256) + 3 * ((FAST_SH * 32 - 64 + 255) / 256));   // (>= FAST_SH * FAST_SW; see the shares below)
    __shared__ uint16_t queue[FAST_PQ_TOTAL];
    __shared__ uint16_t gqueue[(FAST_QR0 + FAST_QR0 + FAST_QR0 + FAST_QR0) * 32];    // groups of 4 positions that failed the cheap reject (row * 32 + group), a quarter per wave
    __shared__ uint32_t qcnt[4], gcnt[4];
    // survivors of the block's tiles are staged in LDS and appended to the level's candidate list with ONE returning
    // global atomic per flush (a flush per tile kept every tile waiting for its own round trip); same for the histogram
    __shared__ uint32_t stage[FAST_STAGE_CAP], shist[256], nstage, stage_end, stage_base;
    const int f = blockIdx.y;
    const uint8_t* frame_pyr = pyr + (int64_t)f * g.frame_bytes;
    const int t = g.fast_thr;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: row loops and queue bases are wave-uniform)
    const int first = blockIdx.x * FAST_TPB;
    FastTile T = fast_tile_geo(tile_tab, first);
    uint32_t pre[FAST_NLD], rel0 = 0;
    int rel_level = -1;
    fast_issue_loads(g, T, frame_pyr, pre, rel0, rel_level);
    shist[threadIdx.x] = 0;
    if (threadIdx.x == 0) { nstage = 0; stage_end = 0xFFFFFFFFu; }
    int stage_level = T.l;
    auto flush = [&](int lvl) {                                // called by the whole block, after a barrier
        const uint32_t ns = min(min(nstage, (uint32_t)FAST_STAGE_CAP), stage_end);   // reservations are monotonic: the staged ones are a prefix
        if (threadIdx.x == 0 && ns) stage_base = atomicAdd(cand_count + (size_t)f * g.nlevels + lvl, ns);
        __syncthreads();
        if (ns) {
            const LevelGeom& Ls = g.lv[lvl];
            uint32_t* clist = cand + (size_t)f * g.cand_per_frame + Ls.cand_ofs;
            const uint32_t base = stage_base;
            for (uint32_t i = threadIdx.x; i < ns; i += 256)
                if (base + i < (uint32_t)Ls.cand_cap) clist[base + i] = stage[i];
            const uint32_t hv = shist[threadIdx.x];
            if (hv) atomicAdd(hist + ((size_t)f * g.nlevels + lvl) * 256 + threadIdx.x, hv);
            shist[threadIdx.x] = 0;
        }
        __syncthreads();
        if (threadIdx.x == 0) { nstage = 0; stage_end = 0xFFFFFFFFu; }
    };
    for (int it = 0; it < FAST_TPB && first + it < g.fast_tiles; ++it) {
    if (it) {
        __syncthreads();                                       // the previous tile's readers of raw / sc / queue are done
        if (T.l != stage_level || nstage > (uint32_t)FAST_STAGE_CAP / 2) flush(stage_level);
        stage_level = T.l;
    }
    if ((int)threadIdx.x < FAST_LT) {
#pragma unroll
        for (int k = 0; k < FAST_NLD; ++k) {
            const int i = (int)threadIdx.x + FAST_LT * k;
            if (i < FAST_NDW * FAST_RH) reinterpret_cast<uint32_t*>(&raw[0][0])[i] = pre[k];
        }
    }
    for (int i = threadIdx.x; i < FAST_SW * FAST_SH / 16; i += 256) reinterpret_cast<uint4*>(&sc[0][0])[i] = make_uint4(0, 0, 0, 0);
    // next tile's pixels (always issued — past the end the last tile is re-read and dropped — so that `pre` stays a
    // plain register array)
    const FastTile Tn = fast_tile_geo(tile_tab, min(first + it + 1, g.fast_tiles - 1));
    fast_issue_loads(g, Tn, frame_pyr, pre, rel0, rel_level);
    const int l = T.l;
    const LevelGeom& L = g.lv[l];
    const int x0 = T.x0, y0 = T.y0;
    constexpr int xoff = FAST_XO - 3;
    __syncthreads();
    // Phase A — quick reject over every score position (x0-1+sx, y0-1+sy); wave w takes score rows w, w+4, ...  Every arc of 9
    // contains one pixel of each antipodal pair, so a pair whose two pixels are both within t of the centre rules the pixel out;
    // the four pairs tested (A1: a cheap sufficient form of the vertical one, on everything; A2: all four exactly, on what is
    // left) leave only corner-like pixels, which are queued for the full test.  Positions past the keep-region's 1-px halo are
    // not needed (score 0); needed ones are >= 3 px inside the level.  Every wave appends to its own quarters of the two queues
    // with wave-uniform counts: no atomics, no waits.
#if defined(FAST_ABL) && FAST_ABL == 0      /* timing experiments only (results invalid): 0 = load + commit only, 1 = + A1, 2 = + A2, 3 = + B */
    T = Tn; continue;
#endif
    const fast_us2 tt = {(unsigned short)t, (unsigned short)t};
    uint32_t myn = 0;
    // the position queue's share of a wave = what it can append in phase A2: four positions per group of the pool it walks
    // (groups 64 w + 256 i + lane of at most FAST_SH * 32): 5 rounds for wave 0, 4 for the others — together every position
    constexpr int FAST_PQ0 = 4 * 64 * ((FAST_SH * 32 + 255) / 256), FAST_PQ1 = 4 * 64 * ((FAST_SH * 32 - 64 + 255) / 256),
                  FAST_PQ2 = 4 * 64 * ((FAST_SH * 32 - 128 + 255) / 256), FAST_PQ3 = 4 * 64 * ((FAST_SH * 32 - 192 + 255) / 256);
    static_assert(FAST_PQ0 + FAST_PQ1 + FAST_PQ2 + FAST_PQ3 <= FAST_PQ_TOTAL, "position queue");
    const int qbase = wave == 0 ? 0 : wave == 1 ? FAST_PQ0 : wave == 2 ? FAST_PQ0 + FAST_PQ1 : FAST_PQ0 + FAST_PQ1 + FAST_PQ2;
    uint16_t* const myq = queue + qbase;
    auto push1 = [&](bool cond, uint32_t val) {
        const uint64_t mk = __builtin_amdgcn_ballot_w64(cond);
        if (mk == 0ull) return;                                          // wave-uniform
        if (cond) myq[myn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u))] = (uint16_t)val;
        myn += (uint32_t)__popcll(mk);
    };
    // A1 — only the sufficient reject, FOUR positions per lane (lanes 0..31: positions 4j .. 4j + 3 of score row sy, lanes 32..63:
    // of row sy + 4): centre, top (-3 rows) and bottom (+3 rows) pixels are one aligned ds_read_b32 each, and
    // max(sum |top - centre|, sum |bottom - centre|) <= t (two v_sad_u8) means no position of the group passes the vertical pair.
    // A group that fails it is queued AS A GROUP (one wave-uniform append per row pair instead of the exact vertical test on the
    // whole wave and four appends: on a text frame a quarter of the waves hold a flagged lane, a tenth of the lanes are flagged).
    uint32_t myg = 0;
    uint16_t* const mygq = gqueue + (wave == 0 ? 0 : wave == 1 ? FAST_QR0 : wave == 2 ? FAST_QR0 + FAST_QR1 : FAST_QR0 + FAST_QR1 + FAST_QR2) * 32;
    const uint32_t* const rawdw = reinterpret_cast<const uint32_t*>(&raw[0][0]) + (FAST_XO >> 2);
    {
        const int half = lane >> 5, j = lane & 31;
        const bool colok = L.rx1 - (x0 - 1) - 4 * j + 1 > 0;              // the group's first position is inside the keep-region's halo
        for (int sy = wave; sy < FAST_SH; sy += 8) {
            if (y0 - 1 + sy > L.ry1) break;
            const int syl = min(sy + 4 * half, FAST_SH - 1);                // (a clamped second row repeats work, never queues)
            const bool rowok = sy + 4 * half < FAST_SH && y0 - 1 + syl <= L.ry1;
            const uint32_t* const c32 = rawdw + (syl + 3) * FAST_NDW + j;
            const uint32_t vc = c32[0], vt = c32[-3 * FAST_NDW], vb = c32[3 * FAST_NDW];
            const uint32_t sad = max(__builtin_amdgcn_sad_u8(vt, vc, 0u), __builtin_amdgcn_sad_u8(vb, vc, 0u));
            const bool flag = rowok && colok && sad > (uint32_t)t;
            const uint64_t mk = __builtin_amdgcn_ballot_w64(flag);
            if (mk == 0ull) continue;
            if (flag) mygq[myg + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u))] = (uint16_t)(syl * 32 + j);
            myg += (uint32_t)__popcll(mk);
        }
    }
    // A2 — all four antipodal pairs on the queued groups, a group per lane, by the wave that queued it, four positions at once as
    // packed u16 (even and odd bytes): with lo = sat(v - t), hi = v + t a pair differs iff min(a, b) < lo or max(a, b) > hi, i.e.
    // max(sat(lo - min), sat(max - hi)) != 0; a position stays iff that holds for every pair (the minimum over the pairs).  The
    // pixels of a pair for the four positions are bytes of the aligned dwords left of, at and right of the group in rows 0, +-2,
    // +-3: eleven ds_read_b32, one v_perm_b32 per (pixel, parity).  Survivors are appended one by one to the share of the position
    // queue of the wave that tested them.
    // The groups of the four waves are pooled (one barrier): a tile of a text frame queues ~110 groups, 27 per wave — four rounds
    // of this loop at 43 % of the lanes when every wave walks its own share, two when the block walks the pool.
    if (lane == 0) gcnt[wave] = myg;
    __syncthreads();
    const uint32_t gc1 = gcnt[0], gc2 = gc1 + gcnt[1], gc3 = gc2 + gcnt[2], ng = gc3 + gcnt[3];
    for (uint32_t k0 = 64u * (uint32_t)wave; k0 < ng; k0 += 256) {
        const uint32_t kq = min(k0 + lane, ng - 1u);
        const bool act = k0 + lane < ng;
        const uint32_t r = (kq >= gc1) + (kq >= gc2) + (kq >= gc3);
        const uint32_t e = gqueue[(r == 0 ? 0u : r == 1 ? (uint32_t)FAST_QR0 * 32 - gc1 : r == 2 ? (uint32_t)(FAST_QR0 + FAST_QR1) * 32 - gc2
                                                                                           : (uint32_t)(FAST_QR0 + FAST_QR1 + FAST_QR2) * 32 - gc3) + kq];
        const int syl = (int)(e >> 5), j = (int)(e & 31u);
        const uint32_t* const c32 = rawdw + (syl + 3) * FAST_NDW + j;
        const uint32_t c = c32[0], pv = c32[-1], nx = c32[1], vt = c32[-3 * FAST_NDW], vb = c32[3 * FAST_NDW];
        const uint32_t pc = c32[2 * FAST_NDW], pp = c32[2 * FAST_NDW - 1], pn = c32[2 * FAST_NDW + 1];
        const uint32_t mc = c32[-2 * FAST_NDW], mp = c32[-2 * FAST_NDW - 1], mn = c32[-2 * FAST_NDW + 1];
        auto pk = [](uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_bit_cast(fast_us2, __builtin_amdgcn_perm(hi, lo, sel)); };
        auto zero_or = [&](fast_us2 v, fast_us2 a, fast_us2 b, fast_us2 lo, fast_us2 hi) {      // != 0 in a half iff a or b is outside [lo, hi] there
            return __builtin_elementwise_max(__builtin_elementwise_sub_sat(lo, __builtin_elementwise_min(a, b)),
                                             __builtin_elementwise_sub_sat(__builtin_elementwise_max(a, b), hi));
        };
        uint32_t keep[2];
#pragma unroll
        for (int par = 0; par < 2; ++par) {                                  // positions 0, 2 (even bytes) / 1, 3 (odd bytes) of the group
            // byte of position k (k = par, par + 2) in a (hi : lo) dword pair at index k + d: selector bytes (k0 + d, 0, k0 + 2 + d, 0)
            const uint32_t s0 = par ? 0x0c030c01u : 0x0c020c00u;            // d = 0:  the position itself (within `c`, `vt`, `vb`)
            const uint32_t s1 = par ? 0x0c040c02u : 0x0c030c01u;            // d = 1
            const uint32_t s2 = par ? 0x0c050c03u : 0x0c040c02u;            // d = 2
            const uint32_t s3 = par ? 0x0c060c04u : 0x0c050c03u;            // d = 3
            const fast_us2 v = pk(0u, c, s0);
            const fast_us2 lo = __builtin_elementwise_sub_sat(v, tt), hi = v + tt;
            fast_us2 d = zero_or(v, pk(0u, vt, s0), pk(0u, vb, s0), lo, hi);                                   // (0, -3) / (0, +3)
            d = __builtin_elementwise_min(d, zero_or(v, pk(nx, c, s3), pk(c, pv, s1), lo, hi));                // (+3, 0) / (-3, 0): bytes k + 3 of (nx : c), k + 1 of (c : pv)
            d = __builtin_elementwise_min(d, zero_or(v, pk(pn, pc, s2), pk(mc, mp, s2), lo, hi));              // (+2, +2) / (-2, -2)
            d = __builtin_elementwise_min(d, zero_or(v, pk(mn, mc, s2), pk(pc, pp, s2), lo, hi));              // (+2, -2) / (-2, +2)
            keep[par] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(d, (fast_us2){1, 1}));          // 0 / 1 per half
        }
        // fl bit k = position k stays: keep[0] = (k0 | k2 << 16), keep[1] = (k1 | k3 << 16)
        const uint32_t m = keep[0] | (keep[1] << 1);
        uint32_t fl = (m | (m >> 14)) & 0xFu;
        const int nin = min(max(L.rx1 - (x0 - 1) - 4 * j + 1, 0), 4);       // positions of the group inside the keep-region's halo
        fl = act ? (fl & (nin >= 4 ? 0xFu : ((1u << nin) - 1u))) : 0u;
        if (__builtin_amdgcn_ballot_w64(fl != 0u) == 0ull) continue;
        const uint32_t p0 = (uint32_t)(syl * FAST_SW + 4 * j);
        push1((fl & 1u) != 0u, p0);
        push1((fl & 2u) != 0u, p0 + 1);
        push1((fl & 4u) != 0u, p0 + 2);
        push1((fl & 8u) != 0u, p0 + 3);
    }
    const uint32_t mym = myn;
#if defined(FAST_ABL) && FAST_ABL == 2
    if (mym == 0xFFFFFFFFu) sc[0][0] = 1;
    T = Tn; continue;
#endif
    if (lane == 0) qcnt[wave] = mym;
    __syncthreads();
    // the queue = the four quarters back to back
    const uint32_t qc1 = qcnt[0], qc2 = qc1 + qcnt[1], qc3 = qc2 + qcnt[2];
    auto qat = [&](uint32_t kq) -> int {
        const uint32_t r = (kq >= qc1) + (kq >= qc2) + (kq >= qc3);
        return queue[(r == 0 ? 0u : r == 1 ? (uint32_t)FAST_PQ0 - qc1 : r == 2 ? (uint32_t)(FAST_PQ0 + FAST_PQ1) - qc2
                                                                            : (uint32_t)(FAST_PQ0 + FAST_PQ1 + FAST_PQ2) - qc3) + kq];
    };
    // Phase B — segment test + cornerScore<16> on the queued positions only
    const uint32_t nqueued = qc3 + qcnt[3];
    for (uint32_t kq = threadIdx.x; kq < nqueued; kq += 256) {
        const int i = qat(kq);
        const int sy = i >> 7, sx = i & 127;
        int score = 0;
        {
            const int cx = sx + 3 + xoff, cy = sy + 3;   // raw coords
            const int v = raw[cy][cx];
            int p[16];
            p[0] = raw[cy + 3][cx];     p[1] = raw[cy + 3][cx + 1]; p[2] = raw[cy + 2][cx + 2];
            p[3] = raw[cy + 1][cx + 3]; p[4] = raw[cy][cx + 3];     p[5] = raw[cy - 1][cx + 3];
            p[6] = raw[cy - 2][cx + 2]; p[7] = raw[cy - 3][cx + 1]; p[8] = raw[cy - 3][cx];
            p[9] = raw[cy - 3][cx - 1]; p[10] = raw[cy - 2][cx - 2]; p[11] = raw[cy - 1][cx - 3];
            p[12] = raw[cy][cx - 3];    p[13] = raw[cy + 1][cx - 3]; p[14] = raw[cy + 2][cx - 2];
            p[15] = raw[cy + 3][cx - 1];
            // cornerScore<16>: max over the 16 arcs of 9 of min(d) and of min(-d), d[k] = v - p[k]; the pixel is a
            // corner iff that maximum exceeds t (an arc of 9 all brighter / all darker by more than t), and then its
            // score is the maximum - 1.  Two arcs per instruction: P[k] = (d[k], d[k+8]) as packed i16, the circular
            // shifts by 8 are half swaps.
            fast_s2 P[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) P[k] = fast_s2{(short)(v - p[k]), (short)(v - p[k + 8])};
            fast_s2 n1[8], x1[8], n2[8], x2[8], n3[8], x3[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {                       // pairs: d[k], d[k+1]
                const fast_s2 o = k < 7 ? P[k + 1] : fast_swap(P[0]);
                n1[k] = __builtin_elementwise_min(P[k], o); x1[k] = __builtin_elementwise_max(P[k], o);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {                       // 4 elements from k
                const fast_s2 on = k < 6 ? n1[k + 2] : fast_swap(n1[k - 6]), ox = k < 6 ? x1[k + 2] : fast_swap(x1[k - 6]);
                n2[k] = __builtin_elementwise_min(n1[k], on); x2[k] = __builtin_elementwise_max(x1[k], ox);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {                       // 8 elements from k
                const fast_s2 on = k < 4 ? n2[k + 4] : fast_swap(n2[k - 4]), ox = k < 4 ? x2[k + 4] : fast_swap(x2[k - 4]);
                n3[k] = __builtin_elementwise_min(n2[k], on); x3[k] = __builtin_elementwise_max(x2[k], ox);
            }
            fast_s2 bestv = {0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k) {                       // the 9th element of arcs k and k+8: d[k+8], d[k]
                const fast_s2 o = fast_swap(P[k]);
                const fast_s2 a9 = __builtin_elementwise_min(n3[k], o), b9 = __builtin_elementwise_max(x3[k], o);
                bestv = __builtin_elementwise_max(bestv, __builtin_elementwise_max(a9, (fast_s2){0, 0} - b9));
            }
            const int best = max((int)bestv.x, (int)bestv.y);
            score = best > t ? best - 1 : 0;
        }
        sc[sy][sx] = (uint8_t)score;      // every other position keeps score 0
    }
#if defined(FAST_ABL) && FAST_ABL == 3
    T = Tn; continue;
#endif
    __syncthreads();
    // Phase C — NMS (strictly greater than all 8 neighbours) + emit, again only over the queue: a survivor is a
    // queued corner inside the output tile and the keep-region.
    uint32_t* ccount = cand_count + (size_t)f * g.nlevels + l;
    uint32_t* clist = cand + (size_t)f * g.cand_per_frame + L.cand_ofs;
    uint32_t* h = hist + ((size_t)f * g.nlevels + l) * 256;
    for (uint32_t k0 = 0; k0 < nqueued; k0 += 256) {
        const uint32_t kq = k0 + threadIdx.x;
        bool keep = false;
        int s = 0, gx = 0, gy = 0;
        if (kq < nqueued) {
            const int i = qat(kq);
            const int sy = i >> 7, sx = i & 127;
            gx = x0 - 1 + sx; gy = y0 - 1 + sy;
            if (sx >= 1 && sx <= FAST_TW && sy >= 1 && sy <= FAST_TH && gx < L.rx1 && gy < L.ry1) {
                const uint8_t* c = &sc[sy][sx];
                s = c[0];
                int m = max(max((int)c[-FAST_SW - 1], (int)c[-FAST_SW]), (int)c[-FAST_SW + 1]);
                m = max(max(m, (int)c[-1]), (int)c[1]);
                m = max(max(m, (int)c[FAST_SW - 1]), max((int)c[FAST_SW], (int)c[FAST_SW + 1]));
                keep = s > m;                                        // s > m >= 0 implies s > 0
            }
        }
        const uint64_t msk = __builtin_amdgcn_ballot_w64(keep);
        if (msk) {
            const uint32_t cnt = (uint32_t)__popcll(msk);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&nstage, cnt);             // LDS
            base = __shfl(base, 0);
            const uint32_t entry = ((uint32_t)s << 24) | ((uint32_t)gy << 12) | (uint32_t)gx;
            if (base + cnt <= (uint32_t)FAST_STAGE_CAP) {
                if (keep) { stage[base + (uint32_t)__popcll(msk & ((1ull << lane) - 1ull))] = entry; atomicAdd(&shist[s], 1u); }
            } else {                                                   // staging full (pathological tile): straight to the list
                if (lane == 0) { atomicMin(&stage_end, base); base = atomicAdd(ccount, cnt); }
                base = __shfl(base, 0);
                if (keep) {
                    uint32_t pos = base + (uint32_t)__popcll(msk & ((1ull << lane) - 1ull));
                    if (pos < (uint32_t)L.cand_cap) clist[pos] = entry;
                    atomicAdd(h + s, 1u);
                }
            }
        }
    }
    T = Tn;
    }   // tile loop
    __syncthreads();
    flush(stage_level);
}

// ---------------------------------------------------------------------------
// [OCV A.6] GaussianBlur 7x7 sigma 2, 8-bit fixed point, BORDER_REFLECT_101:
// out = (sum_j k_j * sum_i k_i * p + 2^15) >> 16.  grid (blur_tiles, B), block 256.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

// Register sliding-window form: a wave owns a strip of 64 dwords (256 columns, of which the inner
// BLUR_TW = 248 are outputs) and BLUR_RH output rows; each lane streams one aligned dword per source row,
// runs the vertical 7-tap pass on packed u16 pairs (v_pk_mad_u16) over a 7-row register ring, fetches
// its neighbours' vertical sums with wave shuffles for the horizontal pass, and stores one dword.
// No LDS, no barriers.  The sum is exact integer arithmetic, so vertical-then-horizontal equals
// OpenCV's horizontal-then-vertical order bit for bit.
typedef unsigned short blur_us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t blur_load_dword(const uint8_t* __restrict__ img, int pitch, int w, int gy, int xs) {
    const uint8_t* row = img + (int64_t)gy * pitch;
    if (xs >= 0 && xs + 3 < w) return *reinterpret_cast<const uint32_t*>(row + xs);
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) d |= (uint32_t)row[reflect101(xs + i, w)] << (8 * i);
    return d;
}

__global__ __launch_bounds__(256) void blur_kernel(PyrGeom g, const uint8_t* __restrict__ pyr,
                                                   uint8_t* __restrict__ blur, OrbTables const* __restrict__ tab) {
    const int f = blockIdx.y;
    const int l = level_of_tile(g, blockIdx.x, true);
    const LevelGeom L = g.lv[l];
    const int tile = blockIdx.x - L.btile0;
    const int ty = tile / L.btx, tx = tile - ty * L.btx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = tx * BLUR_TW;                                   // first output column of the strip
    const int y0 = ty * BLUR_TH + wave * BLUR_RH;                  // first output row of this wave
    if (y0 >= L.h) return;
    const int xs = x0 - 4 + 4 * lane;                              // this lane's 4 columns (lane 0 / 63 = halo)
    const uint8_t* img = pyr + (int64_t)f * g.frame_bytes + L.ofs;
    uint8_t* out = blur + (int64_t)f * g.frame_bytes + L.ofs;
    uint32_t k[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) k[i] = (uint32_t)tab->gk[i];
    uint32_t kk[7];                                                // tap replicated in both halves for packed math
#pragma unroll
    for (int i = 0; i < 7; ++i) kk[i] = k[i] | (k[i] << 16);
    // tap pairs for the horizontal pass: (lo, hi) multiply the (lo, hi) column of a packed pair
    const uint32_t k01 = k[0] | (k[1] << 16), k12 = k[1] | (k[2] << 16), k23 = k[2] | (k[3] << 16), k34 = k[3] | (k[4] << 16),
                   k45 = k[4] | (k[5] << 16), k56 = k[5] | (k[6] << 16), k_0 = k[0] << 16, k6_ = k[6];
    // ring of the last 7 source rows, each as two packed-u16 dwords: A = (p0,p1), B = (p2,p3)
    uint32_t ra[7], rb[7];
    auto load = [&](int y) { return blur_load_dword(img, L.pitch, L.w, reflect101(y, L.h), xs); };
    auto unpack = [&](uint32_t d, uint32_t& A, uint32_t& B) {
        A = __builtin_amdgcn_perm(0u, d, 0x0C010C00u);             // bytes: [p0, 0, p1, 0]
        B = __builtin_amdgcn_perm(0u, d, 0x0C030C02u);             // bytes: [p2, 0, p3, 0]
    };
#pragma unroll
    for (int j = 0; j < 6; ++j) unpack(load(y0 - 3 + j), ra[j], rb[j]);
    const bool inner = lane >= 1 && lane <= 62;
    uint32_t dn0 = load(y0 + 3), dn1 = load(y0 + 4), dn2 = load(y0 + 5);   // newest rows, 3 rows ahead (bandwidth = bytes in flight / latency)
    for (int gI = 0; gI < BLUR_RH / 7; ++gI) {
#pragma unroll
        for (int ph = 0; ph < 7; ++ph) {
            const int i = gI * 7 + ph;                              // output row y0 + i; taps are rows y0+i-3+j in ring[(ph+j)%7]
            unpack(dn0, ra[(ph + 6) % 7], rb[(ph + 6) % 7]);
            dn0 = dn1; dn1 = dn2;
            dn2 = load(min(y0 + i + 6, y0 + BLUR_RH + 2));          // (no row past the strip's last tap is fetched)
            uint32_t va = 0, vb = 0;                                // packed vertical sums (<= 65280 each)
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                blur_us2 ka = __builtin_bit_cast(blur_us2, kk[j]);
                va = __builtin_bit_cast(uint32_t, (blur_us2)(__builtin_bit_cast(blur_us2, ra[(ph + j) % 7]) * ka + __builtin_bit_cast(blur_us2, va)));
                vb = __builtin_bit_cast(uint32_t, (blur_us2)(__builtin_bit_cast(blur_us2, rb[(ph + j) % 7]) * ka + __builtin_bit_cast(blur_us2, vb)));
            }
            // neighbours: left lane's (v1 | v2,v3), right lane's (v0,v1 | v2)
            const uint32_t la = __shfl_up(va, 1), lb = __shfl_up(vb, 1);
            const uint32_t rA = __shfl_down(va, 1), rB = __shfl_down(vb, 1);
            // horizontal pass on the packed pairs, two taps per v_dot2_u32_u16 (sums < 2^24, exact in u32):
            //   la = (x-1, x0)  lb = (x1, x2)  va = (x3, x4)  vb = (x5, x6)  rA = (x7, x8)  rB = (x9, x10),  x_i = column xs-3+i
            //   out0 = k0 x0 + (k1,k2).lb + (k3,k4).va + (k5,k6).vb        out1 = (k0,k1).lb + (k2,k3).va + (k4,k5).vb + k6 x7
            //   out2 = k0 x2 + (k1,k2).va + (k3,k4).vb + (k5,k6).rA        out3 = (k0,k1).va + (k2,k3).vb + (k4,k5).rA + k6 x9
            const uint32_t a0 = dot2_u16(vb, k56, dot2_u16(va, k34, dot2_u16(lb, k12, dot2_u16(la, k_0, 1u << 15))));
            const uint32_t a1 = dot2_u16(vb, k45, dot2_u16(va, k23, dot2_u16(lb, k01, dot2_u16(rA, k6_, 1u << 15))));
            const uint32_t a2 = dot2_u16(rA, k56, dot2_u16(vb, k34, dot2_u16(va, k12, dot2_u16(lb, k_0, 1u << 15))));
            const uint32_t a3 = dot2_u16(rA, k45, dot2_u16(vb, k23, dot2_u16(va, k01, dot2_u16(rB, k6_, 1u << 15))));
            // result byte = bits 16..23 of each sum (clamped: the taps of blur variant 2 sum to 257, so a sum can pass 255 << 16)
            const uint32_t o = __builtin_amdgcn_perm(__builtin_amdgcn_perm(min(a3, 0xFFFFFFu), min(a2, 0xFFFFFFu), 0x0c0c0602u),
                                                     __builtin_amdgcn_perm(min(a1, 0xFFFFFFu), min(a0, 0xFFFFFFu), 0x0c0c0602u), 0x05040100u);
            const int gy = y0 + i;
            if (inner && gy < L.h && xs < L.w) {
                uint8_t* d = out + (int64_t)gy * L.pitch + xs;
                if (xs + 3 < L.w) *reinterpret_cast<uint32_t*>(d) = o;
                else for (int c = 0; c < 4 && xs + c < L.w; ++c) d[c] = (uint8_t)(o >> (8 * c));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// [OCV A.6], slideo_ocv_variants.blur 0 / 1: the blur ORB actually gets for a pyramid level (a SUBMATRIX, so GaussianBlur
// falls through to sepFilter2D with the CV_32F kernel): f32 row pass  s = k0 p0; s += k_i p_i (i = 1..6), f32 column pass
// s = k3 c; s += k_(3+j) (r_(+j) + r_(-j)) (j = 1..3), saturate_cast<uchar>(cvRound(s)).  FMA = products contracted as a
// stock build's AVX2-dispatched filter code has them (variant 0), or separate multiplies and adds (variant 1).
// Same strip geometry as blur_kernel: a lane streams one aligned dword (4 pixels) per source row, gets its neighbours'
// dwords by wave shuffle for the 7-tap row pass, and keeps the last 7 row-filtered rows (4 floats each) in a register ring.
// grid (blur_tiles, B), block 256.
// ---------------------------------------------------------------------------
typedef float blur_f2 __attribute__((ext_vector_type(2)));

// Packed form.  The kernel is bound by VALU issue (PMC, r02: SQ_ACTIVE_INST_VALU ~ all of its SIMD time at about 144
// instructions per 4-pixel row in the first, scalar version), so the arithmetic is organised for v_pk_*_f32 with operand pairs
// that are BORN aligned: a lane processes TWO source rows at a time and pairs every pixel with the pixel below it,
//     P[m] = (p_row a [m], p_row b [m]),  m = 0 .. 9     (20 v_cvt_f32_ubyte, each writing one half of a pair)
// so the row pass of both rows is 4 outputs x 7 packed instructions, and no pair ever has to be re-assembled with moves (pairing
// along x needs every pixel in two differently aligned pairs).  Row-pass results live as row pairs A[k] = (r_2k, r_2k+1) plus
// the interleaved pairs M[k] = (r_2k+1, r_2k+2) (one move per pixel and row pair); the column pass of the output rows (2t, 2t+1)
//     k3 A[t] + k4 (M[t] + M[t-1]) + k5 (A[t+1] + A[t-1]) + k6 (M[t+1] + M[t-2])
// is 7 packed instructions per pixel pair, in OpenCV's order of operations component by component.  The final
// saturate_cast<uchar>(cvRound(s)) is one v_cvt_pk_u8_f32 per pixel (round to nearest even, saturating: tools/cvt_pk_probe.hip).
#ifdef BLUR_VGPRS          /* experiment hook: hard register cap (units of two) */
#define BLUR_KERNEL_ATTR __attribute__((amdgpu_num_vgpr(BLUR_VGPRS)))
#else
#define BLUR_KERNEL_ATTR
#endif
template <bool FMA>
__global__ BLUR_KERNEL_ATTR __launch_bounds__(256) void blur_f32_kernel(PyrGeom g, const uint8_t* __restrict__ pyr,
                                                       uint8_t* __restrict__ blur, OrbTables const* __restrict__ tab,
                                                       const uint8_t* __restrict__ strip_mask) {
    static_assert(BLUR_RH % 8 == 0, "four row pairs per unrolled round");
    SLIDEO_ORB_PRIO();
    const int f = blockIdx.y;
    // strip_mask (frame path): one byte per (frame, tile, wave) strip, set by blur_mark_kernel iff some keypoint's BRIEF
    // samples can fall into the strip; the other strips of the blurred pyramid are never read, so they are not computed.
    // null (pyramid tap): every strip.
    if (strip_mask && !strip_mask[((size_t)f * g.blur_tiles + blockIdx.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)]) return;
    const int l = level_of_tile(g, blockIdx.x, true);
    const LevelGeom L = g.lv[l];
    const int tile = blockIdx.x - L.btile0;
    const int ty = tile / L.btx, tx = tile - ty * L.btx;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the row arithmetic stays on the SALU)
    const int x0 = tx * BLUR_TW;
    const int y0 = ty * BLUR_TH + wave * BLUR_RH;
    if (y0 >= L.h) return;
    const int xs = x0 - 4 + 4 * lane;
    const uint8_t* img = pyr + (int64_t)f * g.frame_bytes + L.ofs;
    uint8_t* out = blur + (int64_t)f * g.frame_bytes + L.ofs;
    blur_f2 kk[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { const float k = tab->gkf[i]; kk[i] = blur_f2{k, k}; }
    auto mad2 = [](blur_f2 a, blur_f2 b, blur_f2 c) -> blur_f2 { return FMA ? __builtin_elementwise_fma(a, b, c) : a * b + c; };
    // A strip whose 4-byte groups all lie inside the level loads one aligned dword per lane and row.  The level's first / last
    // strip (a quarter to two thirds of the strips, depending on the level) has lanes whose group crosses the border.  Four
    // consecutive columns reflect (BORDER_REFLECT_101) onto at most four consecutive bytes, i.e. into two neighbouring aligned
    // dwords: per strip every lane works out ONCE the column of that pair and a v_perm_b32 selector, and per row it loads the
    // two dwords and permutes — no branch, no per-byte loop.  (Reflecting inside the row loop, a while loop per byte executed
    // by the whole wave, made an edge strip cost several times a plain one: 288 VALU instructions per round over the whole
    // launch against 120 for the plain loop.)
    const bool plain = x0 >= 4 && x0 + 4 * 63 <= L.w;              // wave-uniform
    int xa = xs, xb = xs;
    uint32_t sel = 0x03020100u;
    if (!plain) {
        int r[4], lo = INT_MAX;
#pragma unroll
        for (int c = 0; c < 4; ++c) { r[c] = reflect101(xs + c, L.w); lo = min(lo, r[c]); }
        xa = min(lo & ~3, max(L.pitch - 8, 0));
        xb = min(xa + 4, L.pitch - 4);
        sel = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int o = r[c] - xa;                                // 0..7 (3 < o: in the second dword)
            sel |= (uint32_t)(o < 4 ? o : min(r[c] - xb, 3) + 4) << (8 * c);
        }
    }
    const int hh = L.h;
    auto row_of = [&](int y) -> int {                              // BORDER_REFLECT_101 of a row; one fold covers every row an output taps
        if (hh < 8) return reflect101(y, hh);
        int r = y < 0 ? -y : y;
        r = r >= hh ? 2 * (hh - 1) - r : r;
        return min(max(r, 0), hh - 1);                             // (rows past the taps of the last stored output: any valid row)
    };
    // (the permute is applied where the row is consumed, rounds after the loads were issued — not behind a wait right here)
    auto load = [&](int y) -> uint2 {
        const uint8_t* row = img + (int64_t)row_of(y) * L.pitch;
        const uint32_t a = *reinterpret_cast<const uint32_t*>(row + xa);
        return make_uint2(a, *reinterpret_cast<const uint32_t*>(row + xb));     // (plain strip: xb == xa, the same dword again — no branch, no wait)
    };
    auto px4 = [&](uint2 r) -> uint32_t { return __builtin_amdgcn_perm(r.y, r.x, sel); };
    // row pass of source rows ya, ya + 1 -> the lane's 4 outputs of each row, as (row a, row b) pairs
    auto rowpass2 = [&](uint2 ra2, uint2 rb2, blur_f2 (&o)[4]) {
        const uint32_t da = px4(ra2), db = px4(rb2);
        const uint32_t la = __shfl_up(da, 1), ra = __shfl_down(da, 1), lb = __shfl_up(db, 1), rb = __shfl_down(db, 1);
        blur_f2 P[10];
        P[0] = blur_f2{(float)((la >> 8) & 255u), (float)((lb >> 8) & 255u)};
        P[1] = blur_f2{(float)((la >> 16) & 255u), (float)((lb >> 16) & 255u)};
        P[2] = blur_f2{(float)(la >> 24), (float)(lb >> 24)};
        P[3] = blur_f2{(float)(da & 255u), (float)(db & 255u)};
        P[4] = blur_f2{(float)((da >> 8) & 255u), (float)((db >> 8) & 255u)};
        P[5] = blur_f2{(float)((da >> 16) & 255u), (float)((db >> 16) & 255u)};
        P[6] = blur_f2{(float)(da >> 24), (float)(db >> 24)};
        P[7] = blur_f2{(float)(ra & 255u), (float)(rb & 255u)};
        P[8] = blur_f2{(float)((ra >> 8) & 255u), (float)((rb >> 8) & 255u)};
        P[9] = blur_f2{(float)((ra >> 16) & 255u), (float)((rb >> 16) & 255u)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            blur_f2 a = kk[0] * P[j];
#pragma unroll
            for (int i = 1; i < 7; ++i) a = mad2(kk[i], P[j + i], a);
            o[j] = a;
        }
    };
    // Window rows are numbered from y0 - 4: A[k] = window rows (2k, 2k+1), M[k] = (2k+1, 2k+2) = (A[k].y, A[k+1].x), both kept
    // modulo 4.  The output rows (2j, 2j+1) — the first is j = 2 — take A[j-1 .. j+1] and M[j-2 .. j+1], so A[j+2] is made
    // first in every round.  (Window row 0 = y0 - 4 is not a tap of any output; it keeps every pair on an even row.)
    blur_f2 A[4][4], M[4][4];
    auto interleave = [&](int k) {
#pragma unroll
        for (int c = 0; c < 4; ++c) M[k & 3][c] = blur_f2{A[k & 3][c].y, A[(k + 1) & 3][c].x};
    };
    rowpass2(load(y0 - 4), load(y0 - 3), A[0]);
    rowpass2(load(y0 - 2), load(y0 - 1), A[1]);
    rowpass2(load(y0), load(y0 + 1), A[2]);
    rowpass2(load(y0 + 2), load(y0 + 3), A[3]);
    interleave(0); interleave(1); interleave(2);
    const bool inner = lane >= 1 && lane <= 62;
    const int lim = y0 + BLUR_RH + 3;                                  // last source row any output of the strip taps (+ its pair partner)
    // source rows are requested FOUR row pairs (one unrolled round) ahead: 8 loads in flight per lane.  With two pairs ahead
    // the kernel ran at 1.6 TB/s whatever its instruction count (packed or not: 2.18 -> 2.0 ms) — bytes in flight, not VALU.
    uint2 pa[4], pb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { pa[u] = load(min(y0 + 4 + 2 * u, lim)); pb[u] = load(min(y0 + 5 + 2 * u, lim)); }
    for (int gI = 0; gI < BLUR_RH / 8; ++gI) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int t = gI * 4 + ph;                                 // output rows y0 + 2t, y0 + 2t + 1
            constexpr int J0 = 2;
            const int j = ph + J0;                                     // window pair of the outputs, modulo 4 (4 gI drops out)
            rowpass2(pa[ph], pb[ph], A[(j + 2) & 3]);                  // source rows y0 + 2t + 4, + 5
            pa[ph] = load(min(y0 + 2 * t + 12, lim)); pb[ph] = load(min(y0 + 2 * t + 13, lim));    // for round t + 4
            interleave(j + 1);
            uint32_t o0 = 0, o1 = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                blur_f2 a = kk[3] * A[j & 3][c];                      // (k3 c + delta 0 is k3 c in either form)
                a = mad2(kk[4], M[j & 3][c] + M[(j - 1) & 3][c], a);
                a = mad2(kk[5], A[(j + 1) & 3][c] + A[(j - 1) & 3][c], a);
                a = mad2(kk[6], M[(j + 1) & 3][c] + M[(j - 2) & 3][c], a);
                o0 = __builtin_amdgcn_cvt_pk_u8_f32(a.x, (uint32_t)c, o0);
                o1 = __builtin_amdgcn_cvt_pk_u8_f32(a.y, (uint32_t)c, o1);
            }
            const int gy = y0 + 2 * t;
            if (inner && xs < L.w) {
                uint8_t* d = out + (int64_t)gy * L.pitch + xs;
                if (xs + 3 < L.w) {
                    if (gy < L.h) *reinterpret_cast<uint32_t*>(d) = o0;
                    if (gy + 1 < L.h) *reinterpret_cast<uint32_t*>(d + L.pitch) = o1;
                } else {
                    for (int c = 0; c < 4 && xs + c < L.w; ++c) {
                        if (gy < L.h) d[c] = (uint8_t)(o0 >> (8 * c));
                        if (gy + 1 < L.h) d[L.pitch + c] = (uint8_t)(o1 >> (8 * c));
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// [OCV A.4] retainBest(n): threshold = score of the n-th best; every keypoint
// with score >= threshold is kept (ties kept).  grid B, block 64 * nlevels.
// counters layout per frame: see OrbCounters in slideo_capi.hip.
// ---------------------------------------------------------------------------
// kp_cap: keypoints per frame the downstream buffers were sized for when the host does NOT wait for the counts (the
// unit is enqueued in one go); a frame beyond it (ties are kept, so no finite bound is safe) contributes no keypoint and
// raises flag 8, and the host re-runs the unit through the exact-size path.
__global__ void threshold_kernel(PyrGeom g, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ cand_count,
                                 uint32_t* __restrict__ thr, uint32_t* __restrict__ lvl_ofs,
                                 uint32_t* __restrict__ kp_count, uint32_t* __restrict__ flags, uint32_t kp_cap) {
    __shared__ uint32_t kept_s[MAX_LEVELS];
    const int f = blockIdx.x;
    const int l = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t* h = hist + ((size_t)f * g.nlevels + l) * 256;
    // lane i owns bins [4i, 4i+4)
    uint4 b = reinterpret_cast<const uint4*>(h)[lane];
    uint32_t mine = b.x + b.y + b.z + b.w;
    // inclusive suffix sum over lanes (lane 63 = highest scores)
    uint32_t suf = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_down(suf, d);
        if (lane + d < 64) suf += o;
    }
    const uint32_t total = __shfl(suf, 0);
    const uint32_t n = (uint32_t)g.lv[l].quota;
    uint32_t T = 0, kept = total;
    if (total > n) {
        if (n == 0) { T = 256; kept = 0; }
        else {
            uint32_t above = suf - mine;           // count in bins of higher lanes
            bool here = above < n && suf >= n;     // the n-th best lies in my 4 bins
            uint32_t myT = 0, myKept = 0;
            if (here) {
                uint32_t c = above;
                uint32_t bins[4] = {b.x, b.y, b.z, b.w};
                for (int j = 3; j >= 0; --j) {
                    c += bins[j];
                    if (c >= n) { myT = 4 * lane + j; myKept = c; break; }
                }
            }
            uint64_t m = __builtin_amdgcn_ballot_w64(here);
            int src = __ffsll((long long)m) - 1;
            T = __shfl(myT, src); kept = __shfl(myKept, src);
        }
    }
    if (lane == 0) {
        thr[(size_t)f * g.nlevels + l] = T;
        kept_s[l] = kept;
        if (cand_count[(size_t)f * g.nlevels + l] > (uint32_t)g.lv[l].cand_cap) atomicOr(flags, 1u);  // impossible by construction
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t o = 0;
        for (int i = 0; i < g.nlevels; ++i) { lvl_ofs[(size_t)f * g.nlevels + i] = o; o += kept_s[i]; }
        if (o > kp_cap) { atomicOr(flags, 8u); o = 0; }
        kp_count[f] = o;
    }
}

// Exclusive scan of kp_count over frames -> qofs[B+1]; info = {Qtot, max count}.  One block of 1024.
__global__ __launch_bounds__(1024) void scan_kernel(const uint32_t* __restrict__ kp_count, int nframes,
                                                    uint32_t* __restrict__ qofs, uint32_t* __restrict__ info) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t mx[1024];
    const int per = (nframes + 1023) / 1024;
    const int b = threadIdx.x * per, e = min(nframes, b + per);
    uint32_t s = 0, m = 0;
    for (int i = b; i < e; ++i) { s += kp_count[i]; m = max(m, kp_count[i]); }
    part[threadIdx.x] = s; mx[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0, mm = 0;
        for (int i = 0; i < 1024; ++i) { uint32_t v = part[i]; part[i] = run; run += v; mm = max(mm, mx[i]); }
        qofs[nframes] = run; info[0] = run; info[1] = mm;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (int i = b; i < e; ++i) { qofs[i] = run; run += kp_count[i]; }
}

// Candidates with score >= threshold -> items.  item = (level << 24 | y << 12 | x) << 8 | score.
// grid (nlevels, B), block 256.
__global__ __launch_bounds__(256) void compact_kernel(PyrGeom g, const uint32_t* __restrict__ cand,
                                                      const uint32_t* __restrict__ cand_count,
                                                      const uint32_t* __restrict__ thr, const uint32_t* __restrict__ lvl_ofs,
                                                      const uint32_t* __restrict__ qofs, uint32_t* __restrict__ cursor,
                                                      uint64_t* __restrict__ items) {
    const int l = blockIdx.x, f = blockIdx.y;
    const LevelGeom L = g.lv[l];
    const uint32_t n = min(cand_count[(size_t)f * g.nlevels + l], (uint32_t)L.cand_cap);
    const uint32_t T = thr[(size_t)f * g.nlevels + l];
    const uint32_t* clist = cand + (size_t)f * g.cand_per_frame + L.cand_ofs;
    uint32_t* cur = cursor + (size_t)f * g.nlevels + l;
    const uint32_t f_kp = qofs[f + 1] - qofs[f];
    const uint32_t lo = lvl_ofs[(size_t)f * g.nlevels + l];
    uint64_t* dst = items + qofs[f];
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        uint32_t e = clist[i];
        if ((e >> 24) >= T) {
            uint32_t slot = lo + atomicAdd(cur, 1u);
            if (slot < f_kp)
                dst[slot] = ((uint64_t)f << 40) | ((uint64_t)(((uint32_t)l << 24) | (e & 0xFFFFFFu)) << 8) | (e >> 24);   // (frame: constant per sort, spares the consumers a search)
        }
    }
}

// Canonical order (octave, y, x): bitonic sort of each frame's items in LDS.
// grid B, block 1024, dynamic LDS = npow2 * 8 bytes.
__global__ __launch_bounds__(1024) void sort_kernel(const uint32_t* __restrict__ qofs, uint64_t* __restrict__ items, int npow2) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sbuf[];
    const int f = blockIdx.x;
    const uint32_t o = qofs[f], n = qofs[f + 1] - o;
    if (n <= 1 || n > (uint32_t)npow2) return;        // n > npow2 (= KP_SORT_LDS): sort_global_kernel's frame
    int np = 2;
    while ((uint32_t)np < n) np <<= 1;     // np <= npow2
    for (int i = threadIdx.x; i < np; i += 1024) sbuf[i] = (uint32_t)i < n ? items[o + i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= np; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np; i += 1024) {
                int p = i ^ j;
                if (p > i) {
                    uint64_t a = sbuf[i], b = sbuf[p];
                    bool up = (i & k) == 0;
                    if ((a > b) == up) { sbuf[i] = b; sbuf[p] = a; }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < n; i += 1024) items[o + i] = sbuf[i];
}

// Frames with more items than the LDS sort holds (nfeatures in the thousands, or a flood of ties at a retainBest
// threshold): the bitonic network run in place on the frame's items in global memory (they stay in L2).  The variant
// whose comparisons all point the same way — each merge starts with a "flip" (t <-> k-1-t inside a block of k) and
// continues with half-cleaners — so the tail up to the next power of two can stay virtual: a partner index >= n stands
// for +infinity and its compare-exchange is a no-op.  grid B, block 1024; a frame with n <= lds_cap exits at once.
__global__ __launch_bounds__(1024) void sort_global_kernel(const uint32_t* __restrict__ qofs, uint64_t* __restrict__ items, uint32_t lds_cap) {
    const int f = blockIdx.x;
    const uint32_t o = qofs[f], n = qofs[f + 1] - o;
    if (n <= lds_cap) return;
    uint64_t* a = items + o;
    uint32_t np = 2;
    while (np < n) np <<= 1;
    auto cmpx = [&](uint32_t lo, uint32_t hi) {
        if (hi < n) {
            const uint64_t x = a[lo], y = a[hi];
            if (x > y) { a[lo] = y; a[hi] = x; }
        }
    };
    for (uint32_t k = 2; k <= np; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t i = threadIdx.x; i < np / 2; i += 1024) {
            const uint32_t base = (i / hk) * k, t = i % hk;
            cmpx(base + t, base + k - 1 - t);
        }
        __syncthreads();
        for (uint32_t j = hk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < np / 2; i += 1024) {
                const uint32_t lo = (i / j) * 2 * j + (i % j);
                cmpx(lo, lo + j);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// [OCV A.5] ICAngles (fastAtan2: cv_math.hip.h) ; [OCV A.7] rotated BRIEF-256
// One wave per keypoint, 4 keypoints per block.  grid ceil(Qtot/4).
// ---------------------------------------------------------------------------
// Window geometry of describe_kernel (host side: describe_window()).  The rotated pattern reaches
// ceil(half_patch * sqrt 2) pixels from the centre and the 7x7 blur 3 more: R = that + 3.  A wave stages the
// (2R+1) rows x wpd dwords around its keypoint (x aligned down to 4) in LDS, padded to whole wave-instructions.
constexpr int DESC_LDS_ROWS = 4;            // tap rows 0..3 of a sample from LDS, 4..6 from global (measured: 3 -> 1.81, 4 -> 1.53, 5 -> 1.56, 7 -> 1.91 ms)
struct DescWin { int32_t R, wpd, dwords; };
inline DescWin describe_window(int half_patch) {
    DescWin d;
    d.R = (int)ceil((double)half_patch * sqrt(2.0)) + 3;
    d.wpd = ((2 * d.R + 1 + 3) + 3) / 4;
    d.wpd |= 1;                                                  // odd dword pitch: rows start in different banks
    d.dwords = (((2 * d.R + 1) * d.wpd + 2) + 63) & ~63;         // + 2: the 8-byte read of the last tap row may run past the row
    return d;
}

// The blurred level is never materialised for the descriptors: each of the 512 samples of a keypoint needs one value
// of the 7x7 fixed-point Gaussian, 25 k multiply-adds per keypoint against 90 k per keypoint for blurring the whole
// pyramid (the 95 x 95 patches of 1000 keypoints cover more than the pyramid's area, so a sparse blur saves nothing),
// and the full blur's write + read of the pyramid disappears.  Integer-exact: sum_r k_r (sum_j k_j p) is what
// blur_kernel rounds, in any order.  The intensity centroid reads the same window.  (blur_kernel stays for the
// pyramid tap, which is how the parity tests see the blurred levels.)
// One wave per keypoint, 4 keypoints per block, dynamic LDS 4 * dw.dwords * 4 bytes.
__global__ __launch_bounds__(256) void describe_kernel(PyrGeom g, const uint8_t* __restrict__ pyr,
                                                       const OrbTables* __restrict__ tab,
                                                       const uint32_t* __restrict__ qofs, int nframes,
                                                       const uint64_t* __restrict__ items, uint32_t qtot, DescWin dw,
                                                       const uint2* __restrict__ ic_tab, int ic_shift, int ic_entries,
                                                       slideo_keypoint* __restrict__ kp, uint8_t* __restrict__ desc, int atan_fma) {
    extern __shared__ __attribute__((aligned(16))) uint32_t desc_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the keypoint, its level and frame stay on the SALU)
    const uint32_t gi = blockIdx.x * 4 + wave;
    const int lane = threadIdx.x & 63;
    if (qtot == 0xFFFFFFFFu) qtot = qofs[nframes];                  // grid sized by capacity: the count is on the device only
    if (gi >= qtot) return;
    const uint64_t it = items[gi];
    const int f = (int)(it >> 40);                                  // (compact_kernel put the frame there)
    const int score = (int)(it & 255), px = (int)((it >> 8) & 4095), py = (int)((it >> 20) & 4095), l = (int)((it >> 32) & 15);
    const LevelGeom L = g.lv[l];
    const uint8_t* img = pyr + (int64_t)f * g.frame_bytes + L.ofs;
    const int half = g.half_patch;
    const float sf = L.scale;
    const float kx = (float)px * sf, ky = (float)py * sf;
    // computeOrbDescriptors: centre = cvRound(pt * (1/scale)) on the level
    const float inv = 1.f / sf;
    const int cx = (int)rintf(kx * inv), cy = (int)rintf(ky * inv);

    // ---- stage the window: rows cy-R .. cy+R, bytes xa .. xa + 4 wpd (config_supported keeps it inside the level) ----
    const int R = dw.R, wpd = dw.wpd, WP = wpd * 4, wrows = 2 * R + 1;
    uint32_t* win32 = desc_lds + wave * dw.dwords;
    const uint8_t* win = reinterpret_cast<const uint8_t*>(win32);
    const int x0w = cx - R, xa = x0w & ~3, xoff = x0w - xa;
    {
        const uint8_t* src0 = img + (int64_t)(cy - R) * L.pitch + xa;
        const int drow = 64 / wpd, dcol = 64 - drow * wpd;           // flat index + 64 = (row + drow, col + dcol), carry once
        int row = lane / wpd, col = lane - row * wpd;
        for (int i = 0; i < dw.dwords; i += 64) {
            const int rr = min(row, wrows - 1);                       // pad lanes re-read the last row
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + (int64_t)rr * L.pitch + col * 4),
                                             (__attribute__((address_space(3))) void*)(win32 + i), 4, 0, 0);
            row += drow; col += dcol;
            if (col >= wpd) { col -= wpd; ++row; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- intensity centroid over the disc |u| <= umax[|v|] on the unblurred level ----
    // a lane takes 4 pixels (one dword of a disc row) per step; the disc mask and the u weights come as byte weights
    // from ic_tab (geom.h ic_weight_table), so a step is one LDS dword, one table entry, two dot4 and a multiply-add:
    //   m10 = sum (u + half) p - half sum p,   m01 = sum_rows v * (row sum)
    int m10, m01 = 0;
    {
        const int ic0 = (py - cy + R - half) * WP + (px - cx + R - half) + xoff;      // byte of (u, v) = (-half, -half)
        const int ncm = (1 << ic_shift) - 1;
        uint32_t accU = 0, accS = 0;
        for (int e = lane; e < ic_entries; e += 64) {
            const int row = e >> ic_shift, c = e & ncm;
            uint32_t d;
            __builtin_memcpy(&d, win + ic0 + row * WP + 4 * c, 4);     // pad entries read past the disc with weight 0
            const uint2 wgt = ic_tab[e];
            const uint32_t srow = __builtin_amdgcn_udot4(d, wgt.y, 0u, false);
            accU = __builtin_amdgcn_udot4(d, wgt.x, accU, false);
            accS += srow;
            m01 += (row - half) * (int)srow;
        }
        m10 = (int)accU - half * (int)accS;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { m10 += __shfl_xor(m10, d); m01 += __shfl_xor(m01, d); }
    const float angle = fast_atan2f_cv((float)m01, (float)m10, atan_fma != 0);
    const float ang = angle * (float)(3.14159265358979323846 / 180.f);
    const float a = (float)cos((double)ang), b = (float)sin((double)ang);

    // ---- 256 comparisons of blurred samples; the blur is evaluated at the sample: 7 rows x (8 bytes, 2 dot4), 7 mads ----
    const uint32_t g0 = (uint32_t)tab->gk[0], g1 = (uint32_t)tab->gk[1], g2 = (uint32_t)tab->gk[2], g3 = (uint32_t)tab->gk[3],
                   g4 = (uint32_t)tab->gk[4], g5 = (uint32_t)tab->gk[5], g6 = (uint32_t)tab->gk[6];
    const uint32_t k0123 = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24), k456 = g4 | (g5 << 8) | (g6 << 16);
    const uint32_t gr[7] = {g0, g1, g2, g3, g4, g5, g6};
    const int s0 = (R - 3) * WP + (R - 3) + xoff;                     // byte of the top-left tap of the sample at offset (0, 0)
    auto blurred = [&](int sx, int sy) -> uint32_t {
        const uint8_t* pp = win + s0 + sy * WP + sx;
        uint32_t acc = 1u << 15;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            uint64_t v8;
            // random 8-byte gathers are bank-conflict bound in LDS (about 10 % of its peak): the last rows go through
            // the vector L1 instead, where the window's lines sit after the staging pass, and the two queues drain in parallel
            if (r >= DESC_LDS_ROWS) __builtin_memcpy(&v8, img + (int64_t)(cy + sy + r - 3) * L.pitch + (cx + sx - 3), 8);
            else __builtin_memcpy(&v8, pp + r * WP, 8);                // unaligned ds_read_b64
            uint32_t hsum = __builtin_amdgcn_udot4((uint32_t)v8, k0123, 0u, false);
            hsum = __builtin_amdgcn_udot4((uint32_t)(v8 >> 32), k456, hsum, false);
            acc += __umul24(hsum, gr[r]);                             // hsum <= 255 * 256
        }
        return min(acc >> 16, 255u);                                 // (the taps of blur variant 2 sum to 257)
    };
    uint64_t bits[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int pair = lane + 64 * w;          // descriptor bit index
        const uint32_t pw = reinterpret_cast<const uint32_t*>(tab->pattern)[pair];          // (x0, y0, x1, y1) as int8
        const float p0x = (float)(int8_t)(pw & 255), p0y = (float)(int8_t)((pw >> 8) & 255);
        const float p1x = (float)(int8_t)((pw >> 16) & 255), p1y = (float)(int8_t)(pw >> 24);
        const float x0 = p0x * a - p0y * b, y0 = p0x * b + p0y * a;
        const float x1 = p1x * a - p1y * b, y1 = p1x * b + p1y * a;
        const uint32_t t0 = blurred((int)rintf(x0), (int)rintf(y0));
        const uint32_t t1 = blurred((int)rintf(x1), (int)rintf(y1));
        bits[w] = __builtin_amdgcn_ballot_w64(t0 < t1);
    }
    if (lane < 4) reinterpret_cast<uint64_t*>(desc + (size_t)gi * 32)[lane] = bits[lane];
    if (lane == 0) {
        slideo_keypoint k;
        k.x = kx; k.y = ky; k.size = (float)g.patch_size * sf; k.angle = angle;
        k.response = (float)score; k.octave = l;
        kp[gi] = k;
    }
}

// slideo_ocv_variants.blur 0 / 1 (f32 blur): the float sums cannot be evaluated per sample in integer arithmetic, so
// blur_f32_kernel materialises the blurred pyramid and this kernel reads it: the intensity centroid over the disc from the
// UNBLURRED level (dword gathers straight from global memory / L2: one disc row is 16 dwords), then the 512 samples as
// single bytes of the BLURRED level.  One wave per keypoint, 4 keypoints per block, no LDS.  grid ceil(Qtot / 4).
// Which strips of the blurred pyramid will be sampled: one thread per kept keypoint (sorted items), marking every
// (tile, wave) strip of blur_f32_kernel that its sample disc — radius ceil(half_patch sqrt 2) + 1 around the keypoint,
// the reach of the rotated BRIEF pattern after rounding — overlaps.  Keypoints cluster (text, figures), so most strips of a
// frame stay unmarked: 17 % are marked on the benchmark's frames.  qtot == 0xFFFFFFFF: the count is on the device.
__global__ __launch_bounds__(256) void blur_mark_kernel(PyrGeom g, const uint32_t* __restrict__ qofs, int nframes,
                                                        const uint64_t* __restrict__ items, uint32_t qtot, uint8_t* __restrict__ strip_mask) {
    const uint32_t gi = blockIdx.x * 256 + threadIdx.x;
    if (qtot == 0xFFFFFFFFu) qtot = qofs[nframes];
    if (gi >= qtot) return;
    const uint64_t it = items[gi];
    const int lo = (int)(it >> 40);                                 // frame
    const int px = (int)((it >> 8) & 4095), py = (int)((it >> 20) & 4095), l = (int)((it >> 32) & 15);
    const LevelGeom& L = g.lv[l];
    const int R = (int)ceilf((float)g.half_patch * 1.41421357f) + 1;
    const int sx0 = max(px - R, 0) / BLUR_TW, sx1 = min(px + R, L.w - 1) / BLUR_TW;
    const int sy0 = max(py - R, 0) / BLUR_RH, sy1 = min(py + R, L.h - 1) / BLUR_RH;      // strip rows: 4 per tile row
    uint8_t* mk = strip_mask + (size_t)lo * g.blur_tiles * 4;
    for (int sy = sy0; sy <= sy1; ++sy)
        for (int sx = sx0; sx <= sx1; ++sx)
            mk[(size_t)(L.btile0 + (sy >> 2) * L.btx + sx) * 4 + (sy & 3)] = 1;
}

__global__ __launch_bounds__(256) void describe_blurred_kernel(PyrGeom g, const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                               const OrbTables* __restrict__ tab, const uint32_t* __restrict__ qofs, int nframes,
                                                               const uint64_t* __restrict__ items, uint32_t qtot,
                                                               const uint2* __restrict__ ic_tab, int ic_shift, int ic_entries,
                                                               slideo_keypoint* __restrict__ kp, uint8_t* __restrict__ desc, int atan_fma) {
    SLIDEO_ORB_PRIO();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the keypoint, its level and frame stay on the SALU)
    const uint32_t gi = blockIdx.x * 4 + wave;
    const int lane = threadIdx.x & 63;
    if (qtot == 0xFFFFFFFFu) qtot = qofs[nframes];                  // grid sized by capacity: the count is on the device only
    if (gi >= qtot) return;
    const uint64_t it = items[gi];
    const int f = (int)(it >> 40);                                  // (compact_kernel put the frame there)
    const int score = (int)(it & 255), px = (int)((it >> 8) & 4095), py = (int)((it >> 20) & 4095), l = (int)((it >> 32) & 15);
    const LevelGeom L = g.lv[l];
    const uint8_t* img = pyr + (int64_t)f * g.frame_bytes + L.ofs;
    const uint8_t* bimg = blur + (int64_t)f * g.frame_bytes + L.ofs;
    const int half = g.half_patch;
    const float sf = L.scale;
    const float kx = (float)px * sf, ky = (float)py * sf;
    const float inv = 1.f / sf;
    const int cx = (int)rintf(kx * inv), cy = (int)rintf(ky * inv);

    int m10, m01 = 0;
    {
        const uint8_t* ic0 = img + (int64_t)(py - half) * L.pitch + (px - half);      // pixel (u, v) = (-half, -half)
        const int ncm = (1 << ic_shift) - 1;
        uint32_t accU = 0, accS = 0;
        constexpr int ICB = 8;                                       // loads in flight per lane
        for (int e0 = lane; e0 < ic_entries; e0 += 64 * ICB) {
            uint32_t d[ICB]; uint2 wgt[ICB];
#pragma unroll
            for (int u = 0; u < ICB; ++u) {
                const int e = min(e0 + 64 * u, ic_entries - 1);      // (table entries past the disc carry weight 0; the clamp keeps `d` a plain array)
                const int row = e >> ic_shift;
                // rows past the disc (zero-weight pad entries) re-read the disc's last row instead of running off it
                __builtin_memcpy(&d[u], ic0 + (__umul24((uint32_t)min(row, 2 * half), (uint32_t)L.pitch) + 4u * (uint32_t)(e & ncm)), 4);
                wgt[u] = ic_tab[e];
                if (e0 + 64 * u >= ic_entries) wgt[u] = make_uint2(0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < ICB; ++u) {
                const uint32_t srow = __builtin_amdgcn_udot4(d[u], wgt[u].y, 0u, false);
                accU = __builtin_amdgcn_udot4(d[u], wgt[u].x, accU, false);
                accS += srow;
                const int row = min(e0 + 64 * u, ic_entries - 1) >> ic_shift;      // (recomputed: eight registers less than keeping it from the loads)
                m01 += __mul24(row - half, (int)srow);
            }
        }
        m10 = (int)accU - half * (int)accS;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { m10 += __shfl_xor(m10, d); m01 += __shfl_xor(m01, d); }
    const float angle = fast_atan2f_cv((float)m01, (float)m10, atan_fma != 0);
    const float ang = angle * (float)(3.14159265358979323846 / 180.f);
    const float a = (float)cos((double)ang), b = (float)sin((double)ang);

    const uint8_t* bc = bimg + (int64_t)cy * L.pitch + cx;
    uint32_t t0[4], t1[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t pw = reinterpret_cast<const uint32_t*>(tab->pattern)[lane + 64 * w];
        const float p0x = (float)(int8_t)(pw & 255), p0y = (float)(int8_t)((pw >> 8) & 255);
        const float p1x = (float)(int8_t)((pw >> 16) & 255), p1y = (float)(int8_t)(pw >> 24);
        const float x0 = p0x * a - p0y * b, y0 = p0x * b + p0y * a;
        const float x1 = p1x * a - p1y * b, y1 = p1x * b + p1y * a;
        t0[w] = bc[(int64_t)(int)rintf(y0) * L.pitch + (int)rintf(x0)];
        t1[w] = bc[(int64_t)(int)rintf(y1) * L.pitch + (int)rintf(x1)];
    }
    uint64_t bits[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) bits[w] = __builtin_amdgcn_ballot_w64(t0[w] < t1[w]);
    if (lane < 4) reinterpret_cast<uint64_t*>(desc + (size_t)gi * 32)[lane] = bits[lane];
    if (lane == 0) {
        slideo_keypoint k;
        k.x = kx; k.y = ky; k.size = (float)g.patch_size * sf; k.angle = angle;
        k.response = (float)score; k.octave = l;
        kp[gi] = k;
    }
}

}  // namespace slideo
