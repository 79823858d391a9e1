// knn_l2.hip.h — exact squared-L2 k-NN between 128-dimensional u8 descriptors (SIFT-shaped) as an N x M x 128
// integer contraction on the CDNA4 matrix cores.   (SURVEY §8(d) "cfg2", §8(f) N4: a north-star extension — the
// reference has no float-descriptor path, so the parity target is this repository's own CPU restatement
// the brute-force CPU restatement kept with the test infrastructure.)
//
// With every component centred, x' = x - 128 in [-128, 127] (one XOR 0x80 per byte),
//     |q - t|^2 = |q'|^2 + |t'|^2 - 2 <q', t'>
// and <q', t'> is exact in v_mfma_i32_32x32x32_i8: 4 instructions per 32 x 32 tile of pairs, like the Hamming
// engine (knn_tile.hip.h), whose structure this kernel shares: tile-major train operand (a 128-byte row is 8 chunks of
// 16 B, exactly the FP4 layout), 512-query blocks, two 32-query B tiles per wave, LDS ring filled by LDS-DMA and
// guarded by per-slot counters, per-lane thresholds, pushed candidates and batched flushes.  Differences:
//   * the score of a pair is s = 2 <q',t'> - |t'|^2 (larger is nearer; d^2 = |q'|^2 - s).  The train rows are laid out in
//     ascending norm order (a permutation built when the set is prepared), so within a 32-row tile the norms are almost
//     equal and the fast path needs no per-register correction: 2 max(<q',t'>) - (the tile's smallest norm) bounds every s;
//     the exact norms (512 B per super-tile, staged next to it) are only touched in the slow path;
//   * d^2 needs 23 bits and the row 23: keys are 64-bit (d^2 << 32 | row), lists are 32 x u64 per query.
// Ties go to the lower row; pad rows carry a norm of 2^30 and can never qualify.
#pragma once
#include <limits.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knn_tile.hip.h"

namespace slideo {

typedef int knl_v4i __attribute__((ext_vector_type(4)));
typedef int knl_v16i __attribute__((ext_vector_type(16)));

constexpr unsigned long long KNL_EMPTY = ~0ull;
constexpr int KNL_QPB = KT_WAVES * 64;            // queries per block (two 32-query B tiles per wave)
constexpr int KNL_PAD_NORM = 1 << 30;
constexpr int KNL_THR_OPEN = -(1 << 30) + (1 << 24);
constexpr size_t KNL_PEND_WORDS_PER_WAVE = (size_t)2 * KT_PEND_CAP * 64 * 2;      // u64 keys

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
// caller's row perm[i]: ascending |t'|^2 (ties by row) INSIDE a tile, the tiles in a shuffled order (slideo_capi.hip
// l2_prepare), padded to a multiple of KT_ST_ROWS rows;
// neg_norm[i] = -|t'|^2 (pad rows: -2^30, perm -1).  One thread per (sorted row, chunk).
__global__ __launch_bounds__(256) void knl_expand_train_kernel(const uint8_t* __restrict__ t, int nt, int nt_pad,
                                                               const int32_t* __restrict__ perm, const int32_t* __restrict__ norm,
                                                               uint4* __restrict__ out, int32_t* __restrict__ neg_norm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nt_pad * 8) return;
    const int row = i >> 3, c = i & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    const int src = perm[row];                                          // -1: pad row (the tile order is shuffled: pads can sit inside the stream)
    if (src >= 0) {
        v = reinterpret_cast<const uint4*>(t + (size_t)src * 128)[c];
        v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
    }
    out[(size_t)(row >> 5) * 256 + c * 32 + (row & 31)] = v;
    if (c == 0) neg_norm[row] = src >= 0 ? -norm[src] : -KNL_PAD_NORM;
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

// q: [nq][128] u8; tx / tnn: expanded train and negated norms (above); out: [nq][KL] u64 keys (one segment).
// KL = list length: 8 serves k <= 8 (the k = 2 of a ratio test) with 16 list registers and few insertions
// (15.8 ms per 126.6e9 pairs), 16 serves k <= 16 (18.0 ms), 32 serves k <= 32, whose 64 list registers make the
// flush spill (still exact, 26.4 ms).
// Grid ceil(nq / 512), block 512.
template <int KL>
__global__ __launch_bounds__(KT_THREADS, 4) void knn_l2_kernel(const uint8_t* __restrict__ q, int nq,
                                                               const uint4* __restrict__ tx, const int32_t* __restrict__ tnn,
                                                               const int32_t* __restrict__ perm, int nt_pad,
                                                               unsigned long long* __restrict__ out,
                                                               unsigned long long* __restrict__ pend_ws) {
    __shared__ uint4 lds[KT_RING][KT_ST_U4];
    __shared__ __attribute__((aligned(16))) int32_t lds_nn[KT_RING][KT_ST_ROWS];
    __shared__ uint32_t s_filled[KT_RING], s_done[KT_RING];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, ql = lane & 31;
    const int qbase = blockIdx.x * KNL_QPB + wave * 64;
    const int qi = qbase + lane;                                    // the query whose list this lane owns
    const int nst = nt_pad / KT_ST_ROWS;
    unsigned long long* const PA = pend_ws + ((size_t)blockIdx.x * KT_WAVES + wave) * (KNL_PEND_WORDS_PER_WAVE / 2);
    unsigned long long* const PB = PA + (size_t)KT_PEND_CAP * 64;

    // B operands: lane l holds, of query (l & 31) of each tile, the 16 components [32 s + 16 (l >> 5), +16) of k-step s
    knl_v4i bq0[4], bq1[4];
    int nqA = 0, nqB = 0;                                           // |q'|^2 of this lane's tile-0 / tile-1 query
    {
        const uint8_t* q0 = q + (size_t)min(qbase + ql, nq - 1) * 128;
        const uint8_t* q1 = q + (size_t)min(qbase + 32 + ql, nq - 1) * 128;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 w0 = reinterpret_cast<const uint4*>(q0)[2 * s + half], w1 = reinterpret_cast<const uint4*>(q1)[2 * s + half];
            const uint32_t a[4] = {w0.x ^ 0x80808080u, w0.y ^ 0x80808080u, w0.z ^ 0x80808080u, w0.w ^ 0x80808080u};
            const uint32_t b[4] = {w1.x ^ 0x80808080u, w1.y ^ 0x80808080u, w1.z ^ 0x80808080u, w1.w ^ 0x80808080u};
            bq0[s] = knl_v4i{(int)a[0], (int)a[1], (int)a[2], (int)a[3]};
            bq1[s] = knl_v4i{(int)b[0], (int)b[1], (int)b[2], (int)b[3]};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int x = (int)(int8_t)(a[k] >> (8 * c)), y = (int)(int8_t)(b[k] >> (8 * c));
                    nqA += x * x; nqB += y * y;
                }
        }
        nqA += __shfl_xor(nqA, 32); nqB += __shfl_xor(nqB, 32);     // the other half of the components
    }
    uint4* const my_list = reinterpret_cast<uint4*>(out + (size_t)min(qi, nq - 1) * KL);
    const bool owner_valid = qi < nq;
    if (owner_valid) {
#pragma unroll
        for (int i = 0; i < KL / 2; ++i) my_list[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    // a pair qualifies iff s > thr, thr = |q'|^2 - (k-th d^2 of that query).  While the list is not full thr = KNL_THR_OPEN:
    // below every real row's score (>= -3 * 2^21) and above every pad row's (-2^30).
    int thrA = KNL_THR_OPEN, thrB = KNL_THR_OPEN;
    uint32_t cntA = 0, cntB = 0;
    const int nq_own = half ? nqB : nqA;                            // |q'|^2 of the query this lane OWNS (tile = half)

    auto flush = [&]() {
        const uint32_t cA_lo = __shfl(cntA, ql), cA_hi = __shfl(cntA, ql + 32);
        const uint32_t cB_lo = __shfl(cntB, ql), cB_hi = __shfl(cntB, ql + 32);
        const uint32_t c_lo = half ? cB_lo : cA_lo, c_hi = half ? cB_hi : cA_hi;
        const unsigned long long* PP = (half ? PB : PA) + ql;
        unsigned long long lst[KL];
#pragma unroll
        for (int i = 0; i < KL / 2; ++i) {
            const uint4 v = my_list[i];
            lst[2 * i] = ((unsigned long long)v.y << 32) | v.x; lst[2 * i + 1] = ((unsigned long long)v.w << 32) | v.z;
        }
        for (uint32_t base = 0; __builtin_amdgcn_ballot_w64(base < c_lo) != 0ull; base += KT_FLUSH_BATCH) {
            unsigned long long e[KT_FLUSH_BATCH];
#pragma unroll
            for (int i = 0; i < KT_FLUSH_BATCH; ++i) e[i] = base + i < c_lo ? PP[(base + i) * 64] : KNL_EMPTY;
#pragma unroll
            for (int i = 0; i < KT_FLUSH_BATCH; ++i) knl_insert<KL>(lst, e[i]);
        }
        for (uint32_t base = 0; __builtin_amdgcn_ballot_w64(base < c_hi) != 0ull; base += KT_FLUSH_BATCH) {
            unsigned long long e[KT_FLUSH_BATCH];
#pragma unroll
            for (int i = 0; i < KT_FLUSH_BATCH; ++i) e[i] = base + i < c_hi ? PP[(base + i) * 64 + 32] : KNL_EMPTY;
#pragma unroll
            for (int i = 0; i < KT_FLUSH_BATCH; ++i) knl_insert<KL>(lst, e[i]);
        }
        if (owner_valid) {
#pragma unroll
            for (int i = 0; i < KL / 2; ++i)
                my_list[i] = make_uint4((uint32_t)lst[2 * i], (uint32_t)(lst[2 * i] >> 32), (uint32_t)lst[2 * i + 1], (uint32_t)(lst[2 * i + 1] >> 32));
        }
        const int t = lst[KL - 1] == KNL_EMPTY ? KNL_THR_OPEN : nq_own - (int)(lst[KL - 1] >> 32);
        cntA = 0; cntB = 0;
        thrA = __shfl(t, ql);
        thrB = __shfl(t, 32 + ql);
    };

    auto stage = [&](int jj, int sl) {
        constexpr int PER_WAVE = KT_ST_U4 / KT_WAVES;
        const uint4* src = tx + (size_t)jj * KT_ST_U4 + wave * PER_WAVE + lane;
#pragma unroll
        for (int i = 0; i < PER_WAVE / 64; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * i),
                                             (__attribute__((address_space(3))) void*)&lds[sl][wave * PER_WAVE + 64 * i], 16, 0, 0);
    };
    // the super-tile's 128 negated norms: wave 0, 16 B per lane of its first half (published with the wave's `filled` count)
    auto stage_norms = [&](int jj, int sl) {
        if (wave == 0 && lane < KT_ST_ROWS / 4)
            reinterpret_cast<uint4*>(lds_nn[sl])[lane] = reinterpret_cast<const uint4*>(tnn + (size_t)jj * KT_ST_ROWS)[lane];
    };
    auto signal = [&](uint32_t* f) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wait_ge = [&](uint32_t* f, uint32_t target) {
        while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
            __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    if (tid < KT_RING) { s_filled[tid] = 0; s_done[tid] = 0; }
    __syncthreads();
    for (int j = 0; j < KT_AHEAD && j < nst; ++j) { stage(j, j); stage_norms(j, j); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int j = 0; j < KT_AHEAD && j < nst; ++j) signal(&s_filled[j]);
    for (int j = 0; j < nst; ++j) {
        const int slot = j % KT_RING;
        const int jn = j + KT_AHEAD, ns = jn % KT_RING;
        const bool more = jn < nst;
        if (more) {
            wait_ge(&s_done[ns], (uint32_t)KT_WAVES * (uint32_t)(jn / KT_RING));
            stage(jn, ns);
        }
        wait_ge(&s_filled[slot], (uint32_t)KT_WAVES * (uint32_t)(j / KT_RING + 1));
        const uint4* L = lds[slot] + lane;
        uint4 f0 = L[0], f1 = L[64];
#pragma unroll 1
        for (int tile = 0; tile < KT_ST_ROWS / 32; ++tile) {
            knl_v16i a0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0;
            __builtin_amdgcn_s_setprio(KT_MFMA_PRIO);
            {
                const uint4 f2 = L[tile * 256 + 128], f3 = L[tile * 256 + 192];
                const knl_v4i v0 = {(int)f0.x, (int)f0.y, (int)f0.z, (int)f0.w}, v1 = {(int)f1.x, (int)f1.y, (int)f1.z, (int)f1.w};
                a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(v0, bq0[0], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(v0, bq1[0], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(v1, bq0[1], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(v1, bq1[1], a1, 0, 0, 0);
                const knl_v4i v2 = {(int)f2.x, (int)f2.y, (int)f2.z, (int)f2.w}, v3 = {(int)f3.x, (int)f3.y, (int)f3.z, (int)f3.w};
                a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(v2, bq0[2], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(v2, bq1[2], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(v3, bq0[3], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(v3, bq1[3], a1, 0, 0, 0);
                const uint4* Ln = L + min(tile + 1, KT_ST_ROWS / 32 - 1) * 256;
                f0 = Ln[0]; f1 = Ln[64];
            }
            __builtin_amdgcn_s_setprio(0);
            // Fast path.  The score of a pair is s = 2 <q',t'> - |t'|^2.  The train rows are sorted by norm, so the 32 rows
            // of a tile have almost the same |t'|^2 and  s <= 2 max(<q',t'>) + nnmax  with nnmax = the tile's largest
            // negated norm (its first row) is nearly tight: one v_max3 ladder on the raw accumulators and one shift-add
            // per query tile decide whether anything can reach the threshold — no per-register norm correction.
            const int nnmax = lds_nn[slot][tile * 32];
            int ma[5], mb[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                ma[k] = max(max(a0[3 * k], a0[3 * k + 1]), a0[3 * k + 2]);
                mb[k] = max(max(a1[3 * k], a1[3 * k + 1]), a1[3 * k + 2]);
            }
            const int m0 = 2 * max(max(max(a0[15], ma[0]), ma[1]), max(max(ma[2], ma[3]), ma[4])) + nnmax;
            const int m1 = 2 * max(max(max(a1[15], mb[0]), mb[1]), max(max(mb[2], mb[3]), mb[4])) + nnmax;
            if (__builtin_amdgcn_ballot_w64(m0 >= thrA || m1 >= thrB) != 0ull) {
                // Slow path: exact scores; register r of a lane is row (r & 3) + 8 (r >> 2) + 4 half of the tile.  The test is
                // non-strict (d^2 <= k-th d^2): in norm order a later row may have a LOWER original index than the list's
                // k-th entry, and the exact insert decides.
                int ia[16], ib[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int4 nn = *reinterpret_cast<const int4*>(&lds_nn[slot][tile * 32 + 8 * g + 4 * half]);
                    const int n4[4] = {nn.x, nn.y, nn.z, nn.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int x = a0[4 * g + k], y = a1[4 * g + k];
                        ia[4 * g + k] = 2 * x + n4[k]; ib[4 * g + k] = 2 * y + n4[k];
                    }
                }
                const uint32_t row0 = (uint32_t)(j * KT_ST_ROWS + tile * 32 + 4 * half);
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int ga = k < 5 ? max(max(ia[3 * (k < 5 ? k : 0)], ia[3 * (k < 5 ? k : 0) + 1]), ia[3 * (k < 5 ? k : 0) + 2]) : ia[15];
                    const int gb = k < 5 ? max(max(ib[3 * (k < 5 ? k : 0)], ib[3 * (k < 5 ? k : 0) + 1]), ib[3 * (k < 5 ? k : 0) + 2]) : ib[15];
                    if (__builtin_amdgcn_ballot_w64(ga >= thrA || gb >= thrB) == 0ull) continue;
#pragma unroll
                    for (int r = 3 * k; r < 3 * k + 3 && r < 16; ++r) {
                        const uint32_t row = row0 + (r & 3) + 8 * (r >> 2);
                        const bool h0 = ia[r] >= thrA, h1 = ib[r] >= thrB;
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(h0 || h1) != 0ull, 0)) {
                            if (h0 || h1) {
                                const unsigned long long orig = (uint32_t)perm[row];          // row of the caller's train matrix
                                if (h0) { PA[cntA * 64 + lane] = ((unsigned long long)(uint32_t)(nqA - ia[r]) << 32) | orig; ++cntA; }
                                if (h1) { PB[cntB * 64 + lane] = ((unsigned long long)(uint32_t)(nqB - ib[r]) << 32) | orig; ++cntB; }
                            }
                        }
                    }
                }
                if (__builtin_amdgcn_ballot_w64(cntA >= (uint32_t)KT_FLUSH_AT || cntB >= (uint32_t)KT_FLUSH_AT) != 0ull) flush();
            }
        }
        signal(&s_done[slot]);
        if (more) {
            stage_norms(jn, ns);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            signal(&s_filled[ns]);
        }
    }
    flush();
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

}  // namespace slideo
