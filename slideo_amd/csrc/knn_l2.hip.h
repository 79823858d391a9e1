// knn_l2.hip.h — exact squared-L2 k-NN between 128-dimensional u8 descriptors (SIFT-shaped) as an N x M x 128
// integer contraction on the CDNA4 matrix cores.   (SURVEY §8(d) "cfg2", §8(f) N4: a north-star extension — the
// reference has no float-descriptor path, so the parity target is this repository's own CPU restatement
// the brute-force CPU restatement kept with the test infrastructure.)
//
// With every component centred, x' = x - 128 in [-128, 127] (one XOR 0x80 per byte),
//     |q - t|^2 = |q'|^2 + |t'|^2 - 2 <q', t'>
// and <q', t'> is exact in v_mfma_i32_32x32x32_i8: 4 instructions per 32 x 32 tile of pairs.  The search IS the Hamming
// engine's kernel body (knn_tile.hip.h, knn_tile_body) instantiated with another metric (KtL2): same tile-major train operand
// (a 128-byte row is 8 chunks of 16 B, exactly the FP4 layout), same LDS ring, staging, skewed accumulator groups, tile test,
// push / flush protocol.  What the metric changes:
//   * the score of a pair is s = 2 <q',t'> - |t'|^2 (larger is nearer; d^2 = |q'|^2 - s).  The train rows are laid out in
//     ascending norm order inside 32-row tiles (a permutation built when the set is prepared; the tiles themselves are
//     streamed in a shuffled order), so within a tile the norms are almost equal and the fast path needs no per-register
//     correction: 2 max(<q',t'>) - (the tile's smallest norm) bounds every s; the exact norms (the side array staged with
//     the super-tile) are only touched in the slow path;
//   * d^2 needs 23 bits and the row 23: keys are 64-bit (d^2 << 32 | row), lists are KL x u64 per query.
// Ties go to the lower row; pad rows carry a norm of 2^30 and can never qualify.
// This file keeps what is specific to the L2 set: the norms and operand-expansion kernels, the u64 list insert, the kernel
// wrappers and the key unpacking.
#pragma once
#include <limits.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knn_tile.hip.h"

namespace slideo {

// |t'|^2 of every train row (centred components), one thread per row
__global__ __launch_bounds__(256) void knl_norms_kernel(const uint8_t* __restrict__ t, int nt, int32_t* __restrict__ norm) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= nt) return;
    const uint4* p = reinterpret_cast<const uint4*>(t + (size_t)row * 128);
    int ss = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint4 v = p[c];
        const uint32_t w[4] = {v.x ^ 0x80808080u, v.y ^ 0x80808080u, v.z ^ 0x80808080u, v.w ^ 0x80808080u};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b) { const int x = (int)(int8_t)(w[k] >> (8 * b)); ss += x * x; }
    }
    norm[row] = ss;
}

// train [nt][128] u8 -> centred i8, tile-major (tile of 32 rows = [chunk 0..7][row 0..31][16 B]); row i of the stream is the
// caller's row perm[i] (-1: pad row, all zero): ascending |t'|^2 (ties by row) INSIDE a tile, the tiles in a shuffled order
// (slideo_capi.hip l2_prepare), padded to a multiple of KT_ST_ROWS rows.  One thread per (stream row, chunk).
__global__ __launch_bounds__(256) void knl_expand_train_kernel(const uint8_t* __restrict__ t, int nt_pad, const int32_t* __restrict__ perm,
                                                               uint4* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nt_pad * 8) return;
    const int row = i >> 3, c = i & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    const int src = perm[row];
    if (src >= 0) {
        v = reinterpret_cast<const uint4*>(t + (size_t)src * 128)[c];
        v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
    }
    out[(size_t)(row >> 5) * 256 + c * 32 + (row & 31)] = v;
}

template <int KL>
__device__ __forceinline__ void knl_insert(unsigned long long (&lst)[KL], unsigned long long key) {
    // sorted ascending; lst[i] <- max(lst[i-1], min(key, lst[i])), from the top so that old values are still in place
#pragma unroll
    for (int i = KL - 1; i >= 1; --i) {
        const unsigned long long lo = key < lst[i] ? key : lst[i];
        lst[i] = lst[i - 1] > lo ? lst[i - 1] : lo;
    }
    lst[0] = key < lst[0] ? key : lst[0];
}

// The search itself is knn_tile_body (knn_tile.hip.h) with the KtL2 metric: the 2-tile wave shape (4 waves / SIMD, 512-query
// blocks), the whole train set as one segment.  KL = list length: 8 serves k <= 8 (the k = 2 of a ratio test) with 16 list
// registers and few insertions, 16 serves k <= 16, 32 the rest (its 32 x u64 list spills in the flush — still exact).
// q: [nq][128] u8; tx: expanded train; side: [n_st][KT_SIDE_U32] (negated norms as i32 | original rows, pad rows -2^30 / -1);
// tile_norms: [n_st] int4 = the negated norm of the first (smallest-norm) row of each of the super-tile's 4 tiles;
// out: [nq][KL] u64 keys.  Grid ceil(nq / 512), block 512.
// (the k <= 32 instance spills 72 registers in its flush at 4 waves per SIMD; at 2 — -DKNL_K32_WAVES=2: 166 registers, no scratch — the
// launch is 13 % SLOWER, 23.8 against 21.1 ms at the headline pair count: the spill is the cheaper of the two)
#ifndef KNL_K32_WAVES
#define KNL_K32_WAVES 4
#endif
template <int KL>
__global__ __launch_bounds__(KT_THREADS, KL == 32 ? KNL_K32_WAVES : 4) void knn_l2_kernel(const uint8_t* __restrict__ q, int nq, const uint4* __restrict__ tx,
                                                               const uint32_t* __restrict__ side, const uint4* __restrict__ tile_norms,
                                                               int nt_pad, unsigned long long* __restrict__ out,
                                                               unsigned long long* __restrict__ pend_ws, float prune_tol) {
    // prune_tol: 0 = exact lists; > 0 = exact only for the neighbours that can pass sqrt(d) < sqrt(best) * tol (knl_prune_bound)
    knn_tile_body<2, KtL2<KL>>(q, nq, tx, side, tile_norms, nt_pad, nt_pad / KT_ST_ROWS, out, pend_ws, prune_tol, nullptr);
}

// keys [nq][kl] u64 -> idx [nq][k] (-1 = none), dist [nq][k] (squared L2)
__global__ void knl_unpack_kernel(const unsigned long long* __restrict__ keys, int nq, int kl, int k, int32_t* __restrict__ idx,
                                  uint32_t* __restrict__ dist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * k) return;
    const int qi = i / k, r = i - qi * k;
    const unsigned long long key = keys[(size_t)qi * kl + r];
    if (key == KNL_EMPTY) { idx[i] = -1; dist[i] = 0xFFFFFFFFu; }
    else { idx[i] = (int32_t)(key & 0xFFFFFFFFull); dist[i] = (uint32_t)(key >> 32); }
}

// SIFT matcher mode (slideo_matcher_use_sift): Lowe's ratio test on the two nearest rows of the squared-L2 search, written as
// neighbour lists in the HAMMING key format (knn.hip.h: distance << 23 | row, KNN_EMPTY padding) so that the vote kernel — ratio
// branch, ratio 1 — and everything after it run unchanged: entry 0 = (0, nearest row), entry 1 = (pass ? 1 : 0, second row), and
// "0 < 1 * entry 1's distance" is the test's outcome.  pass iff sqrt(d1) < ratio * sqrt(d2) in f32 (BFMatcher returns the square
// roots as floats; IEEE sqrt: the CPU restatement evaluates the same expression).  A query with fewer than two neighbours casts
// no vote (knnMatch with k = 2 needs both).   lists: [nq][kl] u64 ascending (d^2 << 32 | row).  grid ceil(nq / 256).
__global__ __launch_bounds__(256) void l2_ratio_keys_kernel(const unsigned long long* __restrict__ lists, int kl, int nq, float ratio,
                                                            uint32_t* __restrict__ keys, int klist) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const unsigned long long k0 = lists[(size_t)q * kl], k1 = lists[(size_t)q * kl + 1];
    uint32_t o0 = KNN_EMPTY, o1 = KNN_EMPTY;
    if (k0 != KNL_EMPTY) {
        o0 = (uint32_t)k0 & KNN_IDX_MASK;
        if (k1 != KNL_EMPTY) {
            const float d1 = sqrtf((float)(uint32_t)(k0 >> 32)), d2 = sqrtf((float)(uint32_t)(k1 >> 32));
            o1 = ((d1 < ratio * d2 ? 1u : 0u) << KNN_KEY_SHIFT) | ((uint32_t)k1 & KNN_IDX_MASK);
        }
    }
    uint4* out = reinterpret_cast<uint4*>(keys + (size_t)q * klist);
    out[0] = make_uint4(o0, o1, KNN_EMPTY, KNN_EMPTY);
    for (int i = 1; i < klist / 4; ++i) out[i] = make_uint4(KNN_EMPTY, KNN_EMPTY, KNN_EMPTY, KNN_EMPTY);
}

// SIFT matcher mode with the PATH'S OWN vote (slideo_matcher_use_sift, ratio 0): the reference's tolerance rule (mo/lib.rs:268-282: a
// neighbour counts iff d < best * tolerance, f32, strict) on the distances BFMatcher(NORM_L2) returns, sqrt(d^2) in f32.  The
// outcome per neighbour goes into the distance field of a Hamming-format list — entry 0: 1 if the nearest row itself passes
// (it does unless its distance is 0: 0 < 0 is false, as in the reference), else 0; entry r: 1 if it passes, else 2 — and the
// vote kernel's tolerance branch, run with tolerance 1.5, reproduces it: limit = field0 * 1.5, a field passes iff it is below.
// lists: [nq][kl] u64 ascending (d^2 << 32 | row); k <= kl <= klist.   grid ceil(nq / 256).
__global__ __launch_bounds__(256) void l2_tol_keys_kernel(const unsigned long long* __restrict__ lists, int kl, int k, int nq, float tol,
                                                          uint32_t* __restrict__ keys, int klist) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const unsigned long long* L = lists + (size_t)q * kl;
    uint32_t* out = keys + (size_t)q * klist;
    const unsigned long long k0 = L[0];
    float lim = 0.f;
    bool p0 = false;
    if (k0 != KNL_EMPTY) { const float d0 = sqrtf((float)(uint32_t)(k0 >> 32)); lim = d0 * tol; p0 = d0 < lim; }
    for (int r = 0; r < klist; ++r) {
        uint32_t o = KNN_EMPTY;
        if (r < k && r < kl) {
            const unsigned long long kr = L[r];
            if (kr != KNL_EMPTY) {
                const bool pass = p0 && sqrtf((float)(uint32_t)(kr >> 32)) < lim;
                o = ((r == 0 ? (p0 ? 1u : 0u) : (pass ? 1u : 2u)) << KNN_KEY_SHIFT) | ((uint32_t)kr & KNN_IDX_MASK);
            }
        }
        out[r] = o;
    }
}

}  // namespace slideo
