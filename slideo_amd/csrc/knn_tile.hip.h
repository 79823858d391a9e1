// knn_tile.hip.h — the exact k-NN engine on the CDNA4 matrix cores: ONE kernel body (knn_tile_body) over the wave shape (NT)
// and the metric — Hamming distance of 256-bit descriptors on {0,1} x {0,1} FP4 operands (KtHamming, the hot path, described
// below) and squared L2 of 128-dimensional u8 descriptors on centred i8 operands (KtL2, knn_l2.hip.h).
//
// Same contract as knn.hip.h (keys = distance << 23 | train_row, the k smallest, ties to the lower row) and the same
// skeleton as the first matrix-core engine (LDS ring filled by LDS-DMA and guarded by per-slot counters, several 32-query
// B tiles per wave held in registers, skewed accumulator groups, pushed candidates + batched flushes into lists that live in
// the output buffer, vote-acceptance bound fused into the threshold), ONE template over the wave shape:
//   NT = 4 query tiles per wave, 2 waves per SIMD, 1024-query blocks, one per CU  (leaves half of every SIMD's registers
//          and the rest of the LDS to the kernels of the other units in flight; the shape for full batches)
//   NT = 2 query tiles per wave, 4 waves per SIMD, 512-query blocks, two per CU    (fills the chip from fewer queries)
//
// What changed against that engine is the operand alphabet.  There every descriptor bit b was 1 - 2b in {+1, -1} and
// <q', t'> = 256 - 2 Hamming.  The chip clocks to its power budget, and what the FP4 multiplier array burns depends on the
// VALUES it is fed: with 16 v_mfma_scale_f32_32x32x64_f8f6f4 (r02's form; the kernel now issues the unscaled encoding, KtHamming::mfma) + 32 VALU per wave-iteration (tools/mfma_operand_power.hip,
// profiles/r02_mfma_operand_power.txt) the chip sustains 7.3 PFLOP/s on +-1 x +-1 operands, 7.3 on {0,1} x +-1, 8.4 on
// {0,1} x {0,1} (three of four products are zero) and 8.75 on all-zero operands.  So both operands are the raw bits,
// b -> b in {0, 1} (FP4 e2m1: 0x0 / 0x2), the contraction is dot = popcount(q & t), and
//       Hamming(q, t) = |q| + |t| - 2 dot.
// |q| is a per-lane constant.  |t| varies per row, which would cost one VALU per accumulator register in the fast path —
// so the train rows are laid out in ascending |t| order (a stable counting sort when the set is prepared; keys carry the
// caller's row through a permutation staged next to the tile; the TILES are then streamed in a fixed pseudo-random order, see
// prepare_train_bits in slideo_capi.hip for why), and a tile of 32 rows has (nearly) one norm: with
// B = the largest distance a query still accepts, a row qualifies iff d <= B  <=>  dot - |t|/2 > h,  h = (|q| - B - 1) / 2,
// and for a whole tile  max(dot) > h + nmin/2  is necessary (nmin = the tile's smallest norm, a scalar) — one v_add per
// query tile and train tile, then the same v_max3 ladder on the raw bit patterns as before (dot >= 0, so the integer
// order of the patterns is the float order, and a negative threshold compares below everything, which is what it means).
// The exact per-row test, the norms and the original row numbers are touched in the slow path only (LDS, staged with
// the tile).  Because norm order is not row order a later row can have a LOWER original index than a list's k-th entry:
// the acceptance test is non-strict (d <= k-th distance) and the sorted insert decides.
#pragma once
#include <limits.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knn.hip.h"

namespace slideo {

typedef int knn_v8i __attribute__((ext_vector_type(8)));
typedef float knn_v16f __attribute__((ext_vector_type(16)));

constexpr int KT_WAVES = 8;                    // waves per block (and the waves of a block that stage the ring, whatever its size)
constexpr int KT_WAVES12 = 12;                 // the three-waves-per-SIMD block (knn_tile2w12_kernel)
constexpr int KT_THREADS = KT_WAVES * 64;
constexpr int KT_ST_ROWS = 128;                // rows per super-tile (the unit of LDS staging)
#ifndef KT_RING_V
#define KT_RING_V 4
#endif
#ifndef KT_AHEAD_V
#define KT_AHEAD_V 2
#endif
constexpr int KT_RING = KT_RING_V;             // LDS ring slots (super-tiles resident per block)
constexpr int KT_AHEAD = KT_AHEAD_V;           // a super-tile is staged this many iterations before it is consumed
constexpr int KT_ST_U4 = KT_ST_ROWS * 128 / 16;  // uint4 per super-tile of operand (1024)
constexpr int KT_SIDE_U32 = 2 * KT_ST_ROWS;    // per super-tile: 128 f32 norms, then 128 i32 original rows
#ifndef KT_FLUSH_AT_V
#define KT_FLUSH_AT_V 16
#endif
constexpr int KT_FLUSH_AT = KT_FLUSH_AT_V;
constexpr int KT_FLUSH_BATCH = 4;
constexpr int KT_TPS = KT_ST_ROWS / 32;        // tiles per super-tile
constexpr int KT_PEND_CAP = (KT_FLUSH_AT - 1 + (KT_ST_ROWS / 32) * 16 + 7) & ~7;   // = 80;               // >= KT_FLUSH_AT - 1 + KT_TPS * 16 (flush test once per super-tile; a lane pushes <= 16 keys per tile and query)
constexpr float KT_PAD_NORM = 1024.f;          // norm of the pad rows: no distance bound (<= 512) admits them

#ifdef KT_PROBE        /* experiment: where a wave's cycles go (s_memtime), summed over all waves of all launches */
__device__ unsigned long long kt_probe[8];
__device__ unsigned int kt_wave[64][8][4];            // per block (the first 64 of a launch) and wave: total, slow, wait_done, wait_filled (x 1024 cycles)
#define KT_T0(v) const unsigned long long v = __builtin_readcyclecounter()
#define KT_T1(i, v) kt_acc[i] += __builtin_readcyclecounter() - v
#else
#define KT_T0(v)
#define KT_T1(i, v)
#endif
// min(x of lane l, x of lane l ^ 32) in every lane, on the VALU: v_permlane32_swap exchanges the upper half of one register with the
// lower half of another, and the minimum of the two results is symmetric.  (__shfl_xor goes through the LDS crossbar: a round trip
// behind the operand reads of every wave of the CU, on the slow path of every pushed row.)
__device__ __forceinline__ float kt_min_halves(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ int kt_min_halves(int x) {
    const auto r = __builtin_amdgcn_permlane32_swap((uint32_t)x, (uint32_t)x, false, false);
    return min((int)r[0], (int)r[1]);
}
// The shader clock the search waves run at, measured where they run (slideo_matcher_read_shader_clock; bench.py roofline.shader_clock_mhz):
// wave 0 of every 8th block adds its s_memtime (shader cycles) and s_memrealtime (100 MHz) deltas over the block's life to clk[0], clk[1]
// and counts itself in clk[2].  clk == null (not profiling): nothing.  Stateless — the start values are SUBTRACTED in memory, nothing
// lives in registers through the kernel (the 2-tile shape has none to spare).
__device__ __forceinline__ void kt_clock(unsigned long long* __restrict__ clk, bool stop) {
    if (clk != nullptr && (blockIdx.x & 7u) == 0u && threadIdx.x == 0) {
        const unsigned long long c = __builtin_readcyclecounter(), w = wall_clock64();
        atomicAdd(clk, stop ? c : 0ull - c);
        atomicAdd(clk + 1, stop ? w : 0ull - w);
        if (stop) atomicAdd(clk + 2, 1ull);
    }
}
// ---- the ring's counters, in inline assembly --------------------------------------------------------------------------------
// The compiler orders EVERY LDS access it emits behind all pending LDS-DMA of the wave (`s_waitcnt vmcnt(0)` in front of the first
// ds instruction after a global_load_lds that it cannot prove disjoint): with the counters read and bumped through builtins every
// staging wave waited, twice per super-tile, for the share it had JUST issued — the landing deadline of a fetch was half a
// super-tile instead of the look-ahead the ring was built for (found round 6 in the ISA: the waits sat in front of the early
// peek and of the signal's ds_add).  Issued from inline assembly the counter accesses carry no such wait; what they need — the
// wave's own LDS reads done before a `done` signal, a peeked value landed before it is tested — is waited for explicitly.
__device__ __forceinline__ uint32_t kt_opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }      // (a value the optimiser cannot trace: what is computed from it stays where it is written)
__device__ __forceinline__ uint32_t kt_lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }      // (low half of a flat LDS address = the LDS offset)
__device__ __forceinline__ uint32_t kt_ring_peek(const uint32_t* p) {                                 // issued, NOT waited for: kt_ring_landed() before the value is used
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(kt_lds_addr(p)) : "memory");
    return r;
}
__device__ __forceinline__ void kt_ring_landed() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t kt_ring_read(const uint32_t* p) {
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(kt_lds_addr(p)) : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}
__device__ __forceinline__ void kt_ring_signal(uint32_t* p, int lane) {     // after everything this wave read from / wrote to the LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(kt_lds_addr(p)), "v"(1u) : "memory");
}
__device__ __forceinline__ void kt_ring_wait_ge(const uint32_t* p, uint32_t target) {
    while (kt_ring_read(p) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
template <int NT, int W = KT_WAVES> constexpr int knn_qpb() { return W * 32 * NT; }                         // queries per block
template <int NT> constexpr size_t knn_pend_words_per_wave() { return (size_t)NT * KT_PEND_CAP * 64; }

// 8 bits -> 8 FP4 nibbles: bit b -> 0x2 (+1.0) if set, 0x0 (0.0) if clear; bit i -> nibble i.
__host__ __device__ __forceinline__ uint32_t fp4_bits8(uint32_t byte) {
    uint32_t y = byte & 0xFFu;
    y = (y | (y << 12)) & 0x000F000Fu;
    y = (y | (y << 6)) & 0x03030303u;
    y = (y | (y << 3)) & 0x11111111u;
    return y << 1;
}

// ---- the two metrics of the engine ---------------------------------------------------------------------------------------
// Everything that is the same for both — the LDS ring and its counters, LDS-DMA staging, the skewed accumulator groups and
// their interleaved max trees, the tile test, the push / flush protocol, capacity-sized grids — lives ONCE in knn_tile_body;
// a metric supplies the operand encoding, the matrix instruction, the score arithmetic and the key format.
typedef int knl_v4i __attribute__((ext_vector_type(4)));
typedef int knl_v16i __attribute__((ext_vector_type(16)));

// Hamming distance of 256-bit descriptors (the hot path): {0,1} FP4 operands, f32 accumulators, u32 keys d << 23 | row,
// lists of 32; a row qualifies iff dot - |t| / 2 > h, h = (|q| - B - 1) / 2 (header).
struct KtNoCtx {};
struct KtHamming {
    typedef knn_v16f Acc; typedef knn_v8i Bop; typedef uint32_t Key; typedef float Thr;
    typedef KtNoCtx Ctx;                                 // per-launch context of the row filter (none)
    static constexpr bool hamming = true, filtered = false;
    static __device__ __forceinline__ bool accept(const Ctx&, uint32_t, int) { return true; }
    static constexpr int KL = 32, QBYTES = 32;
    static __device__ __forceinline__ Acc mfma(Acc acc, const uint4& f, const Bop& b) {
        const knn_v8i v = {(int)f.x, (int)f.y, (int)f.z, (int)f.w, 0, 0, 0, 0};
        // Both scale operands are the constant 0: the compiler then selects v_mfma_f32_32x32x64_f8f6f4 — the 64-bit encoding
        // WITHOUT the v_mfma_ld_scale half of v_mfma_scale_… (and without its two scale registers).  Same function on these operands,
        // bit for bit (tools/mfma_unscaled_check.hip: 4.2 M accumulator values, 0 differences; the unit-scale form 0x7F7F7F7F is
        // -DKT_SCALED), 2 - 3 % faster: search alone 7.41 -> 7.24 ms, headline step 10.92 -> 10.64 ms (profiles/r06_experiments.txt 10).
#ifdef KT_SCALED
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v, b, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
#else
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v, b, acc, 4, 4, 0, 0, 0, 0);
#endif
    }
    static __device__ __forceinline__ int raw(float x) { return __float_as_int(x); }
    // k-step s of one query row: its operand fragment and the contribution to |q|
    static __device__ __forceinline__ void load_b(const uint8_t* qrow, int s, int half, Bop& b, int& norm) {
        const uint32_t w = reinterpret_cast<const uint32_t*>(qrow)[2 * s + half];
        norm += __popc(w);
        b = knn_v8i{(int)fp4_bits8(w), (int)fp4_bits8(w >> 8), (int)fp4_bits8(w >> 16), (int)fp4_bits8(w >> 24), 0, 0, 0, 0};
    }
    static __device__ __forceinline__ Thr open_thr(Thr nq) { return (nq - 513.f) * 0.5f; }       // B = 512: anything
    static __device__ __forceinline__ int tile_thr(Thr h, uint32_t tile_norm_bits) { return __float_as_int(h + __uint_as_float(tile_norm_bits)); }
    static __device__ __forceinline__ void insert(Key (&lst)[KL], Key e) { knn_insert<32>(lst, e); }
};

// squared L2 of 128-dimensional u8 descriptors (BASELINE configs[2]): components centred to i8 (x XOR 0x80),
// v_mfma_i32_32x32x32_i8, score s = 2 <q',t'> - |t'|^2 (larger is nearer, d^2 = |q'|^2 - s), u64 keys d^2 << 32 | row,
// lists of KL; a row qualifies iff s >= thr, thr = |q'|^2 - (k-th d^2)  (non-strict: see the header).  The side array holds
// the NEGATED norms, a tile's bound is its first (smallest-norm) row's.
constexpr int KNL_PAD_NORM = 1 << 30;
constexpr int KNL_THR_OPEN = -(1 << 30) + (1 << 24);        // below every real row's score (>= -3 * 2^21), above every pad row's (-2^30)
constexpr unsigned long long KNL_EMPTY = ~0ull;
template <int KL> __device__ __forceinline__ void knl_insert(unsigned long long (&lst)[KL], unsigned long long key);
// The Hamming engine restricted to the LSH CANDIDATES of each query (slideo_config.matcher 1, knn_lsh.hip.h): a row that passes
// the distance test enters a list only if, in some table, its key differs from the query's in at most `mp` bits.  The test
// sits in the slow path (a few pairs per thousand), so the stream runs at the exact engine's rate; thresholds follow the
// lists, which hold candidates only — the result is the k nearest candidates, what knn_lsh_kernel computes by gathering.
struct KtLshCtx { const uint16_t* rkeys; const uint16_t* qkeys; int32_t ntab, mp; };
struct KtHammingLsh : KtHamming {
    typedef KtLshCtx Ctx;
    static constexpr bool filtered = true;
    static __device__ __forceinline__ bool accept(const Ctx& c, uint32_t row, int q) {
        const uint16_t* rk = c.rkeys + (size_t)row * c.ntab;
        const uint16_t* qk = c.qkeys + (size_t)q * c.ntab;
        bool ok = false;
        for (int e = 0; e < c.ntab; ++e) ok = ok || __popc((uint32_t)rk[e] ^ (uint32_t)qk[e]) <= c.mp;
        return ok;
    }
};

// Fused vote filter of the L2 engine (SIFT matcher mode with the tolerance vote): a neighbour can only count if
// sqrtf(d^2) < sqrtf(best^2) * tol in f32 with the query's FINAL best, which is at most the best seen so far — so
// d^2 <= best^2 tol^2 (1 + 1e-5) + 1 is necessary (the slack covers the f32 roundings of the two roots and the product many
// times over; d^2 < 2^23.1, exact in f32's integer range after the ceil).  tol < 1 keeps everything up to the best itself.
__device__ __forceinline__ int knl_prune_bound(int best_d2, float tol) {
    const float t = fmaxf(tol, 1.f);
    return (int)fminf(ceilf((float)best_d2 * t * t * 1.00001f) + 1.f, 1.0e9f);
}

template <int KL_>
struct KtL2 {
    typedef knl_v16i Acc; typedef knl_v4i Bop; typedef unsigned long long Key; typedef int Thr;
    typedef KtNoCtx Ctx;
    static constexpr bool hamming = false, filtered = false;
    static __device__ __forceinline__ bool accept(const Ctx&, uint32_t, int) { return true; }
    static constexpr int KL = KL_, QBYTES = 128;
    static __device__ __forceinline__ Acc mfma(Acc acc, const uint4& f, const Bop& b) {
        const knl_v4i v = {(int)f.x, (int)f.y, (int)f.z, (int)f.w};
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(v, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ int raw(int x) { return x; }
    static __device__ __forceinline__ void load_b(const uint8_t* qrow, int s, int half, Bop& b, int& norm) {
        const uint4 w = reinterpret_cast<const uint4*>(qrow)[2 * s + half];
        const uint32_t a[4] = {w.x ^ 0x80808080u, w.y ^ 0x80808080u, w.z ^ 0x80808080u, w.w ^ 0x80808080u};
        b = knl_v4i{(int)a[0], (int)a[1], (int)a[2], (int)a[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int x = (int)(int8_t)(a[k] >> (8 * c)); norm += x * x; }
    }
    static __device__ __forceinline__ Thr open_thr(Thr) { return KNL_THR_OPEN; }
    // 2 max(dot) + nnmax >= thr  <=>  max(dot) > ceil((thr - nnmax) / 2) - 1
    static __device__ __forceinline__ int tile_thr(Thr thr, uint32_t tile_norm_bits) { return ((thr - (int)tile_norm_bits + 1) >> 1) - 1; }
    static __device__ __forceinline__ void insert(Key (&lst)[KL], Key e) { knl_insert<KL>(lst, e); }
};

// train [nt][8] u32 (packed) -> FP4 {0,1}, tile-major (tile of 32 rows = [chunk 0..7][row 0..31][16 B]: MFMA k-step s of a wave
// needs chunk 2s + (lane >> 5) of row (lane & 31) == byte s*1024 + lane*16 of the tile, one linear KB per k-step), IN NORM
// ORDER: sorted row i is the caller's row perm[i]; padded with all-zero rows to a multiple of KT_ST_ROWS.  One thread per
// (sorted row, chunk).
__global__ __launch_bounds__(256) void knn_tile_expand_kernel(const uint32_t* __restrict__ t, int nt, int nt_pad,
                                                               const int32_t* __restrict__ perm, uint4* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nt_pad * 8) return;
    const int row = i >> 3, c = i & 7;
    const int src = perm[row];                                           // -1: pad row (all zero bits, norm KT_PAD_NORM)
    const uint32_t w = src >= 0 ? t[(size_t)src * 8 + c] : 0u;
    out[(size_t)(row >> 5) * 256 + c * 32 + (row & 31)] = make_uint4(fp4_bits8(w), fp4_bits8(w >> 8), fp4_bits8(w >> 16), fp4_bits8(w >> 24));
}

// q: [nq][8] u32 packed; tx: expanded train; side: [n_st][KT_SIDE_U32] (f32 norms | original rows, pad rows KT_PAD_NORM / -1);
// nminh: [n_st] float4 = half the smallest norm of each of the super-tile's 4 tiles.
// prune_tol: 0 = exact k-NN lists; > 0 = lists are exact only for the neighbours with d < best * prune_tol (the vote's rule).
// Grid (ceil(nq / knn_qpb<NT>()), nseg), block 512.  Segment s covers super-tiles [s * st_per_seg, ...).  out: [seg][nq][32] keys.
template <int NT, typename M, int W = KT_WAVES>
__device__ __forceinline__ void knn_tile_body(const uint8_t* __restrict__ q, int nq, const uint4* __restrict__ tx,
                                              const uint32_t* __restrict__ side, const uint4* __restrict__ nminh, int nt_pad,
                                              int st_per_seg, typename M::Key* __restrict__ out, typename M::Key* __restrict__ pend_ws,
                                              float prune_tol, const uint32_t* __restrict__ nq_dev, typename M::Ctx ctx = typename M::Ctx(),
                                              unsigned long long* __restrict__ clk = nullptr) {
    typedef typename M::Acc Acc;
    typedef typename M::Key Key;
    typedef typename M::Thr Thr;
    constexpr int KL = M::KL;
    static_assert(NT == 2 || NT == 4, "two accumulator groups of NT / 2");
    // nq_dev != null: the grid was sized by CAPACITY and the query count lives on the device (the host did not wait for the
    // ORB stage's counts); blocks past the last query leave at once — before any barrier, the test is block-uniform
    if (nq_dev) nq = (int)*nq_dev;
    static_assert(W % 4 == 0 && W >= KT_WAVES, "waves w, w + 4, ... share a SIMD; waves 0 .. KT_WAVES - 1 stage the ring");
    if ((int)blockIdx.x * knn_qpb<NT, W>() >= nq) return;
    kt_clock(clk, false);
#ifdef KT_PROBE
    unsigned long long kt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    KT_T0(kt_all);
    const unsigned long long kt_wall = wall_clock64();
#endif
    constexpr int G = NT / 2;                                          // accumulators per skew group
    __shared__ uint4 lds[KT_RING][KT_ST_U4];
    __shared__ __attribute__((aligned(16))) uint32_t lds_side[KT_RING][KT_SIDE_U32];
    __shared__ uint32_t s_filled[KT_RING], s_done[KT_RING];           // waves that wrote / finished reading each slot (monotonic)
    __shared__ Thr s_nq[W][NT][32];                            // |q| (Hamming) / |q'|^2 (L2) of every query of the block (read in the slow path and the flush only)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, ql = lane & 31;
    const int qbase = blockIdx.x * knn_qpb<NT, W>() + wave * 32 * NT;
    const int seg = blockIdx.y;
    const int n_st = nt_pad / KT_ST_ROWS;
    const int st0 = seg * st_per_seg, st1 = min(n_st, st0 + st_per_seg);
    const int nst = st1 - st0;
    Key* const P0 = pend_ws + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * W + wave) * knn_pend_words_per_wave<NT>();      // (in keys)
    auto pend = [&](int i) -> Key* { return P0 + (size_t)i * KT_PEND_CAP * 64; };

    // B operands: lane l holds, of query (l & 31) of each tile, the 32 bits of packed dword 2s + (l >> 5) for k-step s
    // (register budget: the kernel must leave room for two waves of the other units' kernels per SIMD — at 2 waves of 224
    // registers only one 64-register wave fits beside it and the overlapped step lost 4 %; hence |q| in LDS and the four
    // pending counts packed into one register)
    typename M::Bop bq[NT][4];
    auto load_queries = [&]() {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const uint8_t* qp = q + (size_t)min(qbase + 32 * i + ql, nq - 1) * M::QBYTES;
            int pc = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) M::load_b(qp, s, half, bq[i][s], pc);
            const int tot = pc + __shfl_xor(pc, 32);                   // (all lanes: a shuffle under `if (half == 0)` would read inactive lanes)
            if (half == 0) s_nq[wave][i][ql] = (Thr)tot;
        }
    };
    load_queries();
    auto nq_of = [&](int i) -> Thr { return s_nq[wave][i][ql]; };     // (same wave wrote it: no barrier needed, the compiler waits for the DS write)
    // list p of this lane: wave-local query 64 p + lane, i.e. tile 2 p + half, column ql
    auto list_of = [&](int p) -> uint4* { return reinterpret_cast<uint4*>(out + ((size_t)seg * nq + min(qbase + 64 * p + lane, nq - 1)) * KL); };
    constexpr int LIST_U4 = KL * (int)sizeof(Key) / 16;               // uint4 per list
#pragma unroll
    for (int p = 0; p < G; ++p)
        if (qbase + 64 * p + lane < nq) {
            uint4* l = list_of(p);
#pragma unroll
            for (int i = 0; i < LIST_U4; ++i) l[i] = make_uint4(~0u, ~0u, ~0u, ~0u);                    // (the empty key of either format)
        }
    // h = (|q| - B - 1) / 2 with B = the largest distance the query still accepts (512 = anything); a row qualifies iff
    // dot - |t| / 2 > h
    Thr h[NT];                                                         // (L2: the score threshold thr)
    uint32_t cntp = 0;                                                 // keys pending in this lane's private buffers, 8 bits per query tile
#pragma unroll
    for (int i = 0; i < NT; ++i) h[i] = M::open_thr(nq_of(i));
#ifdef KT_EXPERIMENT_NOSLOW          // measurement only (tools/knn_experiments.sh): no row ever qualifies, the pure streaming rate
#pragma unroll
    for (int i = 0; i < NT; ++i) h[i] = (Thr)1e9f;
#endif
    auto cnt_of = [&](uint32_t packed, int i) -> uint32_t { return (packed >> (8 * i)) & 255u; };

    // owners drain the pending buffers of their two source lanes into their sorted lists (which live in `out`)
    auto flush = [&]() {
#pragma unroll
        for (int p = 0; p < G; ++p) {
            const int A = 2 * p, B = 2 * p + 1;
            const uint32_t p_lo = __shfl(cntp, ql), p_hi = __shfl(cntp, ql + 32);
            const uint32_t c_lo = cnt_of(p_lo, half ? B : A), c_hi = cnt_of(p_hi, half ? B : A);
            const Key* PP = (half ? pend(B) : pend(A)) + ql;
            uint4* my_list = list_of(p);
            const bool owner_valid = qbase + 64 * p + lane < nq;
            Key lst[KL];
#pragma unroll
            for (int i = 0; i < LIST_U4; ++i) {
                const uint4 v = my_list[i];
                if constexpr (M::hamming) { lst[4 * i] = v.x; lst[4 * i + 1] = v.y; lst[4 * i + 2] = v.z; lst[4 * i + 3] = v.w; }
                else { lst[2 * i] = ((Key)v.y << 32) | v.x; lst[2 * i + 1] = ((Key)v.w << 32) | v.z; }
            }
            // pending keys are fetched KT_FLUSH_BATCH at a time (their L2 latency is paid once per batch); a slot past a
            // lane's count reads as the empty key, whose insertion is a no-op
            for (uint32_t base = 0; __builtin_amdgcn_ballot_w64(base < c_lo) != 0ull; base += KT_FLUSH_BATCH) {
                Key e[KT_FLUSH_BATCH];
#pragma unroll
                for (int i = 0; i < KT_FLUSH_BATCH; ++i) e[i] = base + i < c_lo ? PP[(base + i) * 64] : (Key)~(Key)0;
#pragma unroll
                for (int i = 0; i < KT_FLUSH_BATCH; ++i) M::insert(lst, e[i]);
            }
            for (uint32_t base = 0; __builtin_amdgcn_ballot_w64(base < c_hi) != 0ull; base += KT_FLUSH_BATCH) {
                Key e[KT_FLUSH_BATCH];
#pragma unroll
                for (int i = 0; i < KT_FLUSH_BATCH; ++i) e[i] = base + i < c_hi ? PP[(base + i) * 64 + 32] : (Key)~(Key)0;
#pragma unroll
                for (int i = 0; i < KT_FLUSH_BATCH; ++i) M::insert(lst, e[i]);
            }
            if (owner_valid) {
#pragma unroll
                for (int i = 0; i < LIST_U4; ++i) {
                    if constexpr (M::hamming) my_list[i] = make_uint4(lst[4 * i], lst[4 * i + 1], lst[4 * i + 2], lst[4 * i + 3]);
                    else my_list[i] = make_uint4((uint32_t)lst[2 * i], (uint32_t)(lst[2 * i] >> 32), (uint32_t)lst[2 * i + 1], (uint32_t)(lst[2 * i + 1] >> 32));
                }
            }
            if constexpr (M::hamming) {
                // new bound of the query this lane OWNS: the k-th distance (inclusive: see the header), and with prune_tol the
                // vote's acceptance bound — a neighbour counts iff (float)d < (float)best * tol (f32, strict), best only
                // decreases, so d <= ceil(best * tol) - 1 is necessary for ever counting
                float bnd = lst[31] == KNN_EMPTY ? 512.f : (float)(lst[31] >> KNN_KEY_SHIFT);
                if (prune_tol > 0.f) bnd = fminf(bnd, ceilf((float)(lst[0] >> KNN_KEY_SHIFT) * prune_tol) - 1.f);     // (empty list: 511 * tol, no bound)
                h[A] = (nq_of(A) - __shfl(bnd, ql) - 1.f) * 0.5f;
                h[B] = (nq_of(B) - __shfl(bnd, 32 + ql) - 1.f) * 0.5f;
            } else {
                // thr = |q'|^2 - (k-th d^2) of the query this lane owns (tile A for the lower half-wave, B for the upper)
                int t = lst[KL - 1] == KNL_EMPTY ? KNL_THR_OPEN : nq_of(half ? B : A) - (int)(lst[KL - 1] >> 32);
                if (prune_tol > 0.f && lst[0] != KNL_EMPTY) t = max(t, nq_of(half ? B : A) - knl_prune_bound((int)(lst[0] >> 32), prune_tol));
                h[A] = __shfl(t, ql);
                h[B] = __shfl(t, 32 + ql);
            }
        }
        cntp = 0;
    };

    // ---- LDS ring (no block barrier in the main loop) --------------------------------------------------------------
    //   s_filled[slot] += 1 by each wave once its share of a super-tile has landed (consume when == 8 * use#)
    //   s_done[slot]   += 1 by each wave once it has finished reading the slot     (overwrite when == 8 * use#)
    // this wave's share of a super-tile: 2 KB of operand (two 1-KiB LDS-DMA instructions) and, waves 0..3, 256 B of the
    // side array; lane i of a DMA instruction lands at the wave-uniform LDS base + (its width) * i
    auto stage = [&](int jj, int sl) {
        constexpr int PER_WAVE = KT_ST_U4 / KT_WAVES;                  // uint4 per wave and super-tile
        static_assert(PER_WAVE % 64 == 0, "whole wave-instructions");
        // a wave-uniform base (scalar registers) + a 32-bit lane offset the optimiser cannot trace back to the lane number: a
        // per-lane 64-bit source pointer hoisted out of the main loop is two vector registers the wave shapes do not have
        const uint32_t lo = kt_opaque((uint32_t)lane);
        const char* sb = reinterpret_cast<const char*>(tx + (size_t)(st0 + jj) * KT_ST_U4 + wave * PER_WAVE);
#pragma unroll
        for (int i = 0; i < PER_WAVE / 64; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + (lo * 16u + 1024u * i)),
                                             (__attribute__((address_space(3))) void*)&lds[sl][wave * PER_WAVE + 64 * i], 16, 0, 0);
        if (wave < KT_SIDE_U32 / 64) {
            const char* ss = reinterpret_cast<const char*>(side + (size_t)(st0 + jj) * KT_SIDE_U32 + wave * 64);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ss + lo * 4u),
                                             (__attribute__((address_space(3))) void*)&lds_side[sl][wave * 64], 4, 0, 0);
        }
    };
#ifdef KT_RING_BUILTIN       /* A/B: the counters through compiler builtins (rounds 1 - 5; see kt_ring_peek) */
    auto signal = [&](uint32_t* f) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wait_ge = [&](uint32_t* f, uint32_t target) {
        while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
            __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
#else
    auto signal = [&](uint32_t* f) { kt_ring_signal(f, lane); };
    auto wait_ge = [&](uint32_t* f, uint32_t target) { kt_ring_wait_ge(f, target); };
#endif
    if (tid < KT_RING) { s_filled[tid] = 0; s_done[tid] = 0; }
    __syncthreads();
    const bool stager = W == KT_WAVES || wave < KT_WAVES;             // (wave-uniform) a block of more than KT_WAVES waves: the others only read the ring
    if (stager) {
        for (int j = 0; j < KT_AHEAD && j < nst; ++j) stage(j, j);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int j = 0; j < KT_AHEAD && j < nst; ++j) signal(&s_filled[j]);
    }
    // e_done / e_filled: the two counters as read one tile EARLIER (counters only grow, so an early value that already meets the
    // target is as good as a fresh one; a wave that tests a fresh read pays the LDS round trip — behind the operand reads of
    // every wave of the CU — twice per super-tile with its matrix pipe idle: a quarter of a wave's time at 2 waves per SIMD)
    auto acquire = [&](int j, uint32_t e_done, uint32_t e_filled) {    // group A is about to read super-tile j
        const int jp = j - 1 + KT_AHEAD, jn = j + KT_AHEAD;
#ifndef KT_RING_BUILTIN
        kt_ring_landed();                                              // (the early peeks, issued a tile ago)
#endif
        if (stager && j > 0 && jp < nst) {                             // publish this wave's share staged at acquire(j - 1)
            KT_T0(t_a);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            KT_T1(1, t_a);
            signal(&s_filled[jp % KT_RING]);
        }
        if (stager && jn < nst) {                                      // the slot was last read for super-tile jn - KT_RING (by all W waves)
            KT_T0(t_b);
            const uint32_t tgt = (uint32_t)W * (uint32_t)(jn / KT_RING);
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)e_done) < tgt) wait_ge(&s_done[jn % KT_RING], tgt);
            KT_T1(2, t_b);
            stage(jn, jn % KT_RING);
        }
        KT_T0(t_c);
        const uint32_t tgf = (uint32_t)KT_WAVES * (uint32_t)(j / KT_RING + 1);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)e_filled) < tgf) wait_ge(&s_filled[j % KT_RING], tgf);
        asm volatile("" ::: "memory");
        KT_T1(3, t_c);
    };
#ifdef KT_RING_BUILTIN
    auto peek = [&](const uint32_t* f) -> uint32_t { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
#else
    auto peek = [&](const uint32_t* f) -> uint32_t { return kt_ring_peek(f); };
#endif
    auto mfma = [&](Acc acc, const uint4& f, const typename M::Bop& b) { return M::mfma(acc, f, b); };
    // maxima of the raw bit patterns: triples {3k, 3k+1, 3k+2}, k = 0..4, register 15 apart
    // (only the overall maximum is kept: the slow path recomputes the triple maxima it gates on — ten registers per skew group
    // that would otherwise stay live from the trees to the tests)
    auto tree = [&](const Acc& acc) -> int {
        int tk[5];
#pragma unroll
        for (int k = 0; k < 5; ++k)
            tk[k] = max(max(M::raw(acc[3 * k]), M::raw(acc[3 * k + 1])), M::raw(acc[3 * k + 2]));
        return max(max(max(M::raw(acc[15]), tk[0]), tk[1]), max(max(tk[2], tk[3]), tk[4]));
    };
    // Slow path (about one tile in twenty): exact per-row test with the rows' own norms; usually ONE value of ONE lane
    // qualifies, so every test is a wave-uniform "nobody" branch that falls through.  Register r of a lane is row
    // (r & 3) + 8 (r >> 2) + 4 half of the tile.
    auto candidates = [&](const Acc& acc, int thri, const uint32_t* sd, int tt, Thr& hh, int qt, Key* P) {
        float dbest = M::hamming ? 1024.f : 2.0e9f;
        const Thr nq_i = nq_of(qt);
        uint32_t c = cnt_of(cntp, qt);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int kk = k < 5 ? k : 0;
            const bool gate = (k < 5 ? max(max(M::raw(acc[3 * kk]), M::raw(acc[3 * kk + 1])), M::raw(acc[3 * kk + 2]))
                                     : M::raw(acc[15])) > thri;
            if (__builtin_amdgcn_ballot_w64(gate) == 0ull) continue;
            // the norms of the triple's rows: three LDS reads in flight, one wait (row by row every test paid its own round trip)
            // ... and their original row numbers with them (six reads in flight, one wait): a gate that opens nearly always ends in a
            // push, and the row number fetched only then was one more LDS round trip — behind the operand reads of every wave of the
            // CU — on the path every record-breaking row takes
            uint32_t nrm3[3], row3[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int r = min(3 * k + u, 15);
                nrm3[u] = sd[tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
                if constexpr (M::hamming) row3[u] = sd[KT_ST_ROWS + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];      // (the L2 engine is at its register limit: it reads the row when it pushes)
            }
#pragma unroll
            for (int r = 3 * k; r < 3 * k + 3 && r < 16; ++r) {
                if constexpr (M::hamming) {
                    const float nrm = __uint_as_float(nrm3[r - 3 * k]);
                    const float v = acc[r];
                    const bool hit = __builtin_fmaf(nrm, -0.5f, v) > hh;              // exact: halves of small integers
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(hit) != 0ull, 0)) {
                        bool take = hit;
                        if constexpr (M::filtered) { if (hit) take = M::accept(ctx, row3[r - 3 * k], min(qbase + 32 * qt + ql, nq - 1)); }
                        if (take) {
                            const float d = nq_i + nrm - 2.f * v;
                            P[c * 64 + lane] = ((uint32_t)(int)d << KNN_KEY_SHIFT) | row3[r - 3 * k];
                            ++c;
                            dbest = fminf(dbest, d);
                        }
                    }
                } else {
                    const int sc = 2 * acc[r] + (int)nrm3[r - 3 * k];                 // s = 2 <q',t'> - |t'|^2 (the side array holds -|t'|^2)
                    const bool hit = sc >= hh;
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(hit) != 0ull, 0)) {
                        if (hit) {
                            P[c * 64 + lane] = ((Key)(uint32_t)(nq_i - sc) << 32) | (Key)sd[KT_ST_ROWS + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
                            ++c;
                            dbest = fminf(dbest, (float)(nq_i - sc));      // (d^2 < 2^24: exact)
                        }
                    }
                }
            }
        }
        cntp = (cntp & ~(255u << (8 * qt))) | (c << (8 * qt));
        if constexpr (!M::hamming) {
            if (prune_tol > 0.f) {                                       // (the same, for the L2 engine: knl_prune_bound)
                int bn = dbest < 1.0e9f ? knl_prune_bound((int)dbest, prune_tol) : 0x3FFFFFFF;
                bn = kt_min_halves(bn);
                if (bn < 0x3FFFFFFF) hh = max(hh, (Thr)(nq_i - bn));
            }
        }
        if constexpr (M::hamming) {
            if (prune_tol > 0.f) {
                // Fused vote filter, applied at once instead of at the next flush: every distance seen so far bounds the
                // query's final best from above, so the acceptance bound of the best row this lane just pushed is already
                // valid, and so is the one its partner lane (the other 16 rows of the same query) derived.
                float bn = ceilf(dbest * prune_tol) - 1.f;
                bn = kt_min_halves(bn);
                hh = fmaxf(hh, (nq_i - bn - 1.f) * 0.5f);
            }
        }
    };

    if (nst > 0) {
        const Acc zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        Acc a[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) a[i] = zero;
        // G accumulator chains through the four k-steps of one A tile
#define KT_MFMAS(GB, f0, f1, f2, f3)                                                                                    \
        _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) a[GB + g_] = mfma(zero, f0, bq[GB + g_][0]);                 \
        _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) a[GB + g_] = mfma(a[GB + g_], f1, bq[GB + g_][1]);           \
        _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) a[GB + g_] = mfma(a[GB + g_], f2, bq[GB + g_][2]);           \
        _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) a[GB + g_] = mfma(a[GB + g_], f3, bq[GB + g_][3]);
        // one MFMA, then two instructions of the other group's max trees, 4 G times over
#define KT_INTERLEAVE                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 4 * G; ++i_) {                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
        // trees + thresholds of group GB for the tile whose half norm is nmh_; then the tests
#define KT_TREES(GB, nmh_)                                                                                             \
        int mx_[G], ti_[G];                                                                                           \
        _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) { mx_[g_] = tree(a[GB + g_]); ti_[g_] = M::tile_thr(h[GB + g_], (nmh_)); }
#define KT_TEST(GB, sd_, tt_)                                                                                          \
        {                                                                                                             \
            bool any_ = false;                                                                                        \
            _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) any_ |= mx_[g_] > ti_[g_];                               \
            if (__builtin_amdgcn_ballot_w64(any_) != 0ull) {                                                          \
                KT_T0(t_s);                                                                                           \
                _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_)                                                      \
                    if (__builtin_amdgcn_ballot_w64(mx_[g_] > ti_[g_]) != 0ull)                                       \
                        candidates(a[GB + g_], ti_[g_], sd_, tt_, h[GB + g_], GB + g_, pend(GB + g_));                \
                KT_T1(4, t_s);                                                                                        \
            }                                                                                                         \
        }
        // One tile (tt = its index in the super-tile, compile time): (c*) = F(t) are live on entry, (n*) = F(t + 1) on exit;
        // group A (accumulators 0 .. G-1) already holds tile t.  Lc / Ln: this lane's fragment pointers into the slot of the
        // current / the next tile's super-tile; sdc: side array of the current slot.
#ifdef KT_N23_EARLY      /* experiment: all four fragments of the next tile requested at the tile's start */
#define KT_EARLY_N23(x) x
#define KT_LATE_N23(x)
#else
#define KT_EARLY_N23(x)
#define KT_LATE_N23(x) x
#endif
#define KT_TILE(tt, c0, c1, c2, c3, n0, n1, n2, n3)                                                                    \
        {                                                                                                             \
            const uint4* Lx_ = (tt) == KT_TPS - 1 ? Ln : Lc + ((tt) + 1) * 256;                                       \
            n0 = Lx_[0]; n1 = Lx_[64]; KT_EARLY_N23(n2 = Lx_[128]; n3 = Lx_[192];)                                   \
            const uint32_t nmh_ = (tt) == 0 ? nm4.x : (tt) == 1 ? nm4.y : (tt) == 2 ? nm4.z : nm4.w;                  \
            {                                                                                                         \
                KT_MFMAS(G, c0, c1, c2, c3)                                                                            \
                KT_TREES(0, nmh_)                                                                                     \
                KT_INTERLEAVE                                                                                         \
                _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) asm volatile("" : "+v"(a[G + g_]));   /* keeps the MFMAs above the branch below */ \
                KT_LATE_N23(n2 = Lx_[128]; n3 = Lx_[192];)                                                            \
                KT_TEST(0, sdc, tt)                                                                                   \
            }                                                                                                         \
            {                                                                                                         \
                KT_MFMAS(0, n0, n1, n2, n3)                                                                            \
                KT_TREES(G, nmh_)                                                                                     \
                KT_INTERLEAVE                                                                                         \
                _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) asm volatile("" : "+v"(a[g_]));                      \
                KT_TEST(G, sdc, tt)                                                                                   \
            }                                                                                                         \
        }
        // Wave priority: the two waves a block has on a SIMD (w and w + 4) lead in turns, one super-tile each.  With equal
        // priorities the arbiter favours the older wave (0 .. 3): it runs ahead until the ring stops it (15 % of its time
        // waiting for a slot, against 5 % for waves 4 .. 7 — per-wave s_memtime sums, profiles/r04_experiments.txt) and the
        // SIMD then runs ONE wave's instruction stream; taking turns keeps both within a super-tile of each other.
        static_assert(KT_WAVES == 8, "waves w and w + 4 (and w + 8) share a SIMD");
        acquire(0, 0u, 0u);
        const uint4* Lc = lds[0] + lane;
        uint4 x0 = Lc[0], x1 = Lc[64], x2 = Lc[128], x3 = Lc[192];                                 // F(t)
        uint4 y0, y1, y2, y3;                                                                      // F(t + 1)
        KT_MFMAS(0, x0, x1, x2, x3)
        // the super-tile's four half norms (wave-uniform address: a scalar load; float / int bit patterns) are fetched one super-tile
        // AHEAD: asked for where they are used, the load sat two MFMAs in front of its `s_waitcnt lgkmcnt(0)` — a scalar-cache round
        // trip in the first chain of every super-tile
        uint4 nm4n = nminh[st0];
#pragma unroll 1
        for (int j = 0; j < nst; ++j) {
            const int slot = j % KT_RING;
            if (W == KT_WAVES ? (((j ^ (wave >> 2)) & 1) != 0) : ((j + (wave >> 2)) % (W / 4) == 0)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
            Lc = lds[slot] + lane;
            const uint32_t* sdc = lds_side[slot];
            const uint4 nm4 = nm4n;
            nm4n = nminh[st0 + min(j + 1, nst - 1)];
            // the tile after the segment's last is the last one again: recomputed into group A, never tested
            const uint4* Ln = j + 1 < nst ? lds[(j + 1) % KT_RING] + lane : Lc + (KT_TPS - 1) * 256;
            KT_TILE(0, x0, x1, x2, x3, y0, y1, y2, y3)
            KT_TILE(1, y0, y1, y2, y3, x0, x1, x2, x3)
            uint32_t e_done = 0, e_filled = 0;                           // (0 never meets a positive target: the fresh read decides)
            if (j + 1 < nst) {
                e_filled = peek(&s_filled[(j + 1) % KT_RING]);
                if (j + 1 + KT_AHEAD < nst) e_done = peek(&s_done[(j + 1 + KT_AHEAD) % KT_RING]);
            }
            KT_TILE(2, x0, x1, x2, x3, y0, y1, y2, y3)
            if (j + 1 < nst) acquire(j + 1, e_done, e_filled);         // group A enters the next super-tile in tile 3
            KT_TILE(3, y0, y1, y2, y3, x0, x1, x2, x3)
            signal(&s_done[slot]);
            bool need = false;
#pragma unroll
            for (int i = 0; i < NT; ++i) need |= cnt_of(cntp, i) >= (uint32_t)KT_FLUSH_AT;
            if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                KT_T0(t_f);
                flush();
                KT_T1(5, t_f);
                // everything the loop carries is rebuilt after the (rare) flush instead of kept alive across it: the flush
                // needs 32 registers for the list, and values live across it would be spilled on every path
                load_queries();
                x0 = Ln[0]; x1 = Ln[64]; x2 = Ln[128]; x3 = Ln[192];
                KT_MFMAS(0, x0, x1, x2, x3)
            }
        }
#undef KT_TILE
#undef KT_EARLY_N23
#undef KT_LATE_N23
#undef KT_TEST
#undef KT_TREES
#undef KT_INTERLEAVE
#undef KT_MFMAS
    }
    flush();
    kt_clock(clk, true);
#ifdef KT_PROBE
    KT_T1(0, kt_all);
    kt_acc[6] = 1; kt_acc[7] = wall_clock64() - kt_wall;
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&kt_probe[i], kt_acc[i]);
    if (lane == 0 && wave < 8 && blockIdx.x >= 100 && blockIdx.x < 164 && blockIdx.y == 0) { unsigned int* w = kt_wave[blockIdx.x - 100][wave]; w[0] = (unsigned)(kt_acc[0] >> 10); w[1] = (unsigned)(kt_acc[4] >> 10); w[2] = (unsigned)(kt_acc[2] >> 10); w[3] = (unsigned)(kt_acc[3] >> 10); }
#endif
}

// The two wave shapes.  KT4_VGPRS caps the 4-tile shape's register allocation: at 2 waves per SIMD the compiler would take up
// to 256 registers, but what it leaves is what the other units' kernels (ORB, verify: 36 - 64 registers per wave) live in.
#ifndef KT4_VGPRS
#define KT4_VGPRS 100      /* the backend doubles the request on gfx90a+ (unified VGPR + AGPR file): 100 -> 200 registers */
#endif
__global__ __attribute__((amdgpu_num_vgpr(KT4_VGPRS))) __launch_bounds__(KT_THREADS, 2)
void knn_tile4_kernel(const uint32_t* __restrict__ q, int nq, const uint4* __restrict__ tx, const uint32_t* __restrict__ side,
                      const float4* __restrict__ nminh, int nt_pad, int st_per_seg, uint32_t* __restrict__ out,
                      uint32_t* __restrict__ pend_ws, float prune_tol, const uint32_t* __restrict__ nq_dev) {
    knn_tile_body<4, KtHamming>(reinterpret_cast<const uint8_t*>(q), nq, tx, side, reinterpret_cast<const uint4*>(nminh), nt_pad, st_per_seg, out, pend_ws,
                                prune_tol, nq_dev);
}
#ifdef KT2_CLOBBER
__global__ __launch_bounds__(KT_THREADS, 2)
#elif defined(KT2_VGPRS)   /* experiments only: a cap below the 128 that 4 waves per SIMD allow */
__global__ __attribute__((amdgpu_num_vgpr(KT2_VGPRS))) __launch_bounds__(KT_THREADS, 4)
#else
__global__ __launch_bounds__(KT_THREADS, 4)
#endif
void knn_tile2_kernel(const uint32_t* __restrict__ q, int nq, const uint4* __restrict__ tx, const uint32_t* __restrict__ side,
                      const float4* __restrict__ nminh, int nt_pad, int st_per_seg, uint32_t* __restrict__ out,
                      uint32_t* __restrict__ pend_ws, float prune_tol, const uint32_t* __restrict__ nq_dev, unsigned long long* __restrict__ clk) {
#ifdef KT2_CLOBBER      /* experiment: the register allocation of a larger wave shape without its code */
    asm volatile("" ::: KT2_CLOBBER);
#endif
    knn_tile_body<2, KtHamming>(reinterpret_cast<const uint8_t*>(q), nq, tx, side, reinterpret_cast<const uint4*>(nminh), nt_pad, st_per_seg, out, pend_ws,
                                prune_tol, nq_dev, KtNoCtx(), clk);
}
// Three waves per SIMD: a block of 12 waves (768 queries), ONE per CU by its registers (3 x 128 per SIMD lane; the other 128 and
// 87 KB of LDS stay free for the other units' kernels).  Waves 0 .. 7 stage the ring as in the 8-wave block, all twelve read it.
__global__ __attribute__((amdgpu_waves_per_eu(4, 4))) __launch_bounds__(KT_WAVES12 * 64)
void knn_tile2w12_kernel(const uint32_t* __restrict__ q, int nq, const uint4* __restrict__ tx, const uint32_t* __restrict__ side,
                         const float4* __restrict__ nminh, int nt_pad, int st_per_seg, uint32_t* __restrict__ out,
                         uint32_t* __restrict__ pend_ws, float prune_tol, const uint32_t* __restrict__ nq_dev, unsigned long long* __restrict__ clk) {
    knn_tile_body<2, KtHamming, KT_WAVES12>(reinterpret_cast<const uint8_t*>(q), nq, tx, side, reinterpret_cast<const uint4*>(nminh), nt_pad, st_per_seg, out, pend_ws,
                                            prune_tol, nq_dev, KtNoCtx(), clk);
}
// the same engine over the LSH candidates only (KtHammingLsh)
__global__ __launch_bounds__(KT_THREADS, 4)
void knn_tile2_lsh_kernel(const uint32_t* __restrict__ q, int nq, const uint4* __restrict__ tx, const uint32_t* __restrict__ side,
                          const float4* __restrict__ nminh, int nt_pad, int st_per_seg, uint32_t* __restrict__ out,
                          uint32_t* __restrict__ pend_ws, float prune_tol, const uint32_t* __restrict__ nq_dev, KtLshCtx ctx) {
    knn_tile_body<2, KtHammingLsh>(reinterpret_cast<const uint8_t*>(q), nq, tx, side, reinterpret_cast<const uint4*>(nminh), nt_pad, st_per_seg, out, pend_ws,
                                   prune_tol, nq_dev, ctx);
}

}  // namespace slideo
