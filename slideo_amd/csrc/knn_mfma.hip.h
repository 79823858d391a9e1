// knn_mfma.hip.h — exact Hamming k-NN on the CDNA4 matrix cores.
//
// Same contract as knn.hip.h (keys = distance << 23 | train_row, k smallest,
// ties to the lower row), different engine.  The all-pairs distance matrix is
// a genuine N x M x 256 contraction: with every descriptor bit b mapped to
// (1 - 2b) in {+1,-1},   <q', t'> = 256 - 2 * Hamming(q, t)   exactly.
// +1/-1 are exact in FP4 (e2m1: 0x2 / 0xA), products are +-1 and a sum of 256 of
// them is exact in the f32 accumulator, so
// v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 x FP4, unit E8M0 scales) yields the
// integer dot products bit-exactly at the FP4 matrix rate: 4 instructions per
// 32 x 32 tile of pairs (vs 16 VALU ops per PAIR for xor + popcount, whose 3-operand
// VOP3 forms issue at ~36 T lane-ops/s on this chip — profiles/r01_valu_op_rates.txt).
//
// Layouts
//   train (A operand): expanded once (finalize) to FP4 and stored tile-major:
//     tile T = rows [32T, 32T+32): 4 KB = [chunk c = 0..7][row r = 0..31][16 B], chunk c = the
//     32 bits of packed dword c of the descriptor as 32 nibbles.  MFMA k-step s of a
//     wave needs chunk 2s + (lane >> 5) of row (lane & 31)  ==  byte  s*1024 + lane*16  of the
//     tile: one linear KB per k-step, so HBM -> LDS is a straight copy and the
//     fragment read is a conflict-free ds_read_b128.
//   queries (B operand): lane l holds query (l & 31), expands packed dword 2s + (l >> 5) in
//     registers at kernel start (16 VGPRs).
//   accumulator: lane l holds column (query) l & 31, rows (r&3) + 8*(r>>2) + 4*(l>>5).
//   top-k: 32 sorted keys in VGPRs of the query's owner lane; candidates reach it through a small
//     pending buffer in LDS (see the kernel comment); train segments are merged by knn_merge_kernel.
//
// Block = 8 waves = 512 queries sharing the A tiles through LDS (super-tiles of 128 rows =
// 16 KB, double buffered, one barrier per super-tile).  Fast path per tile and wave:
// 4 ds_read_b128 + 8 MFMA + max-of-16 twice + 2 compares.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knn.hip.h"

namespace slideo {

typedef int knn_v8i __attribute__((ext_vector_type(8)));
typedef float knn_v16f __attribute__((ext_vector_type(16)));

constexpr int KM_WAVES = 8;                    // waves per block
constexpr int KM_THREADS = KM_WAVES * 64;
constexpr int KM_QPB = KM_WAVES * 64;          // queries per block (two 32-query B tiles per wave)
constexpr int KM_ST_ROWS = 128;                // rows per super-tile (one block barrier per super-tile)
constexpr int KM_ST_U4 = KM_ST_ROWS * 128 / 16;  // uint4 per super-tile (1024)
constexpr int KM_STAGE = KM_ST_U4 / KM_THREADS;  // uint4 staged per thread and super-tile

// 8 bits -> 8 FP4 nibbles: bit b -> 0x2 (+1.0) if 0, 0xA (-1.0) if 1; bit i -> nibble i.
__host__ __device__ __forceinline__ uint32_t fp4_expand8(uint32_t byte) {
    uint32_t y = byte & 0xFFu;
    y = (y | (y << 12)) & 0x000F000Fu;
    y = (y | (y << 6)) & 0x03030303u;
    y = (y | (y << 3)) & 0x11111111u;
    return 0x22222222u | (y << 3);
}

// train [nt][8] u32 (packed) -> FP4 tile-major, padded to a multiple of KM_ST_ROWS rows
// (pad rows are all +1; they are excluded by row index in the kernel).  One thread per (row, chunk).
__global__ __launch_bounds__(256) void knn_expand_train_kernel(const uint32_t* __restrict__ t, int nt, int nt_pad,
                                                               uint4* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nt_pad * 8) return;
    const int row = i >> 3, c = i & 7;
    const uint32_t w = row < nt ? t[(size_t)row * 8 + c] : 0u;
    uint4 v = make_uint4(fp4_expand8(w), fp4_expand8(w >> 8), fp4_expand8(w >> 16), fp4_expand8(w >> 24));
    out[(size_t)(row >> 5) * 256 + c * 32 + (row & 31)] = v;
}

// q: [nq][8] u32 packed; tx: expanded train (see above), nt valid rows, nt_pad padded rows.
// Grid (ceil(nq / 512), nseg), block 512.  Segment s covers super-tiles [s*st_per_seg, ...).
// out: [seg][nq][32] keys.
//
// A wave owns 64 queries as two B tiles (tile 0: queries qbase+0..31, tile 1: qbase+32..63) and issues
// 2 x 4 MFMAs per 32-row train tile (the A fragment is read from LDS once for both).  Lane l holds, for
// query (l&31) of each tile, the rows (r&3) + 8(r>>2) + 4(l>>5).  The sorted top-k list of tile-0 query n
// lives in lane n, that of tile-1 query n in lane 32+n ("owner" lanes).
//
// Epilogue, fast path (no cross-lane traffic): every lane keeps the thresholds of BOTH its queries
// (thrA, thrB) and compares them with the maxima of its two accumulator tiles.
// Slow path: a lane whose accumulator beats the threshold PUSHES the candidate key into the owner's small
// pending buffer in LDS (ds_add_rtn slot + ds_write); nothing is inserted yet.  When some owner has
// KM_FLUSH_AT pending keys the wave flushes: owners insert their pending keys (v_med3 chain), thresholds are
// refreshed and re-broadcast.  This batches the divergent part: an insert costs the whole wave ~35 VALU
// instructions, and without batching it ran for what is usually ONE lane's candidate.
// Exactness: rows are offered in ascending 8-row groups and thresholds only tighten at a flush, so a
// candidate with the k-th distance and a higher row than everything in the list is (correctly) rejected by the
// strict filter, and nothing that belongs to the final top-k is ever filtered out.
constexpr int KM_FLUSH_AT = 16;
constexpr int KM_PEND_CAP = 32;                      // >= KM_FLUSH_AT - 1 + 16 (a lane pushes <= 16 keys per tile and query)
constexpr size_t KM_PEND_WORDS_PER_WAVE = (size_t)2 * KM_PEND_CAP * 64;

// Pending buffers: every lane has a PRIVATE buffer per query tile in a global workspace (L2 resident; the
// count lives in a register, no atomics).  The owner of tile-0 query n (lane n) gathers from lanes n and
// n+32, the owner of tile-1 query n (lane 32+n) likewise.  Layout [wave][tile][slot][lane].
// The sorted list of a query lives in the output buffer and is only brought into registers inside a flush,
// which runs between tiles (accumulators dead) when some lane has >= KM_FLUSH_AT keys pending.
__global__ __launch_bounds__(KM_THREADS, 4) void knn_mfma_kernel(const uint32_t* __restrict__ q, int nq,
                                                                 const uint4* __restrict__ tx, int nt, int nt_pad,
                                                                 int st_per_seg, uint32_t* __restrict__ out,
                                                                 uint32_t* __restrict__ pend_ws) {
    __shared__ uint4 lds[2][KM_ST_U4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, ql = lane & 31;
    const int qbase = blockIdx.x * KM_QPB + wave * 64;
    const int qi = qbase + lane;                                    // the query whose list this lane owns
    const int seg = blockIdx.y;
    const int n_st = nt_pad / KM_ST_ROWS;
    const int st0 = seg * st_per_seg, st1 = min(n_st, st0 + st_per_seg);
    uint32_t* const PA = pend_ws + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * KM_WAVES + wave) * KM_PEND_WORDS_PER_WAVE;
    uint32_t* const PB = PA + (size_t)KM_PEND_CAP * 64;

    knn_v8i bq0[4], bq1[4];
    {
        const uint32_t* q0 = q + (size_t)min(qbase + ql, nq - 1) * 8;
        const uint32_t* q1 = q + (size_t)min(qbase + 32 + ql, nq - 1) * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint32_t w0 = q0[2 * s + half], w1 = q1[2 * s + half];
            bq0[s] = knn_v8i{(int)fp4_expand8(w0), (int)fp4_expand8(w0 >> 8), (int)fp4_expand8(w0 >> 16), (int)fp4_expand8(w0 >> 24), 0, 0, 0, 0};
            bq1[s] = knn_v8i{(int)fp4_expand8(w1), (int)fp4_expand8(w1 >> 8), (int)fp4_expand8(w1 >> 16), (int)fp4_expand8(w1 >> 24), 0, 0, 0, 0};
        }
    }
    uint4* const my_list = reinterpret_cast<uint4*>(out + ((size_t)seg * nq + min(qi, nq - 1)) * 32);
    const bool owner_valid = qi < nq;
    if (owner_valid) {
#pragma unroll
        for (int i = 0; i < 8; ++i) my_list[i] = make_uint4(KNN_EMPTY, KNN_EMPTY, KNN_EMPTY, KNN_EMPTY);
    }
    float thrA = -1024.f, thrB = -1024.f;     // dot > thr  <=>  distance < current k-th distance of that query
    uint32_t cntA = 0, cntB = 0;              // keys pending in this lane's private buffers

    // owners drain the pending buffers of their two source lanes into their sorted list
    auto flush = [&]() {
        const uint32_t cA_lo = __shfl(cntA, ql), cA_hi = __shfl(cntA, ql + 32);
        const uint32_t cB_lo = __shfl(cntB, ql), cB_hi = __shfl(cntB, ql + 32);
        const uint32_t c_lo = half ? cB_lo : cA_lo, c_hi = half ? cB_hi : cA_hi;
        const uint32_t* PP = half ? PB : PA;
        uint32_t lst[32];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 v = my_list[i];
            lst[4 * i] = v.x; lst[4 * i + 1] = v.y; lst[4 * i + 2] = v.z; lst[4 * i + 3] = v.w;
        }
        for (uint32_t sidx = 0; __builtin_amdgcn_ballot_w64(sidx < c_lo) != 0ull; ++sidx)
            knn_insert<32>(lst, sidx < c_lo ? PP[sidx * 64 + ql] : KNN_EMPTY);
        for (uint32_t sidx = 0; __builtin_amdgcn_ballot_w64(sidx < c_hi) != 0ull; ++sidx)
            knn_insert<32>(lst, sidx < c_hi ? PP[sidx * 64 + ql + 32] : KNN_EMPTY);
        cntA = 0; cntB = 0;
        if (owner_valid) {
#pragma unroll
            for (int i = 0; i < 8; ++i) my_list[i] = make_uint4(lst[4 * i], lst[4 * i + 1], lst[4 * i + 2], lst[4 * i + 3]);
        }
        const float t = 256.f - 2.f * (float)(lst[31] >> KNN_KEY_SHIFT);
        thrA = __shfl(t, ql);
        thrB = __shfl(t, 32 + ql);
    };

    if (st0 < st1) {
#pragma unroll
        for (int i = 0; i < KM_STAGE; ++i) lds[0][tid + i * KM_THREADS] = tx[(size_t)st0 * KM_ST_U4 + tid + i * KM_THREADS];
    }
    __syncthreads();
    int cur = 0;
    for (int st = st0; st < st1; ++st) {
        uint4 nx[KM_STAGE];
        const bool more = st + 1 < st1;
        if (more) {
#pragma unroll
            for (int i = 0; i < KM_STAGE; ++i) nx[i] = tx[(size_t)(st + 1) * KM_ST_U4 + tid + i * KM_THREADS];
        }
        const uint4* L = lds[cur];
#pragma unroll 1
        for (int tile = 0; tile < KM_ST_ROWS / 32; ++tile) {
            knn_v16f a0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                uint4 a = L[tile * 256 + s * 64 + lane];
                knn_v8i av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
                a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bq0[s], a0, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bq1[s], a1, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            }
            // per-lane group maxima: group g = registers 4g..4g+3 = rows 8g + 4*half + {0..3}
            float g0[4], g1[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                g0[g4] = fmaxf(fmaxf(a0[4 * g4], a0[4 * g4 + 1]), fmaxf(a0[4 * g4 + 2], a0[4 * g4 + 3]));
                g1[g4] = fmaxf(fmaxf(a1[4 * g4], a1[4 * g4 + 1]), fmaxf(a1[4 * g4 + 2], a1[4 * g4 + 3]));
            }
            const float m0 = fmaxf(fmaxf(g0[0], g0[1]), fmaxf(g0[2], g0[3]));
            const float m1 = fmaxf(fmaxf(g1[0], g1[1]), fmaxf(g1[2], g1[3]));
            if (__builtin_amdgcn_ballot_w64(m0 > thrA || m1 > thrB) != 0ull) {
                const int row0 = st * KM_ST_ROWS + tile * 32 + 4 * half;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {                       // ascending 8-row groups
                    if (__builtin_amdgcn_ballot_w64(g0[g4] > thrA || g1[g4] > thrB) == 0ull) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = row0 + 8 * g4 + j;
                        const float v0 = a0[4 * g4 + j], v1 = a1[4 * g4 + j];
                        if (v0 > thrA && row < nt) {                    // candidate for tile-0 query ql
                            PA[cntA * 64 + lane] = ((uint32_t)(256 - (int)v0) << (KNN_KEY_SHIFT - 1)) | (uint32_t)row;
                            ++cntA;
                        }
                        if (v1 > thrB && row < nt) {                    // candidate for tile-1 query ql
                            PB[cntB * 64 + lane] = ((uint32_t)(256 - (int)v1) << (KNN_KEY_SHIFT - 1)) | (uint32_t)row;
                            ++cntB;
                        }
                    }
                }
                if (__builtin_amdgcn_ballot_w64(cntA >= (uint32_t)KM_FLUSH_AT || cntB >= (uint32_t)KM_FLUSH_AT) != 0ull) flush();
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < KM_STAGE; ++i) lds[cur ^ 1][tid + i * KM_THREADS] = nx[i];
        }
        __syncthreads();
        cur ^= 1;
    }
    flush();
}

}  // namespace slideo
