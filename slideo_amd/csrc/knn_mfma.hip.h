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
//   accumulator: lane l holds column (query) l & 31, rows (r&3) + 8*(r>>2) + 4*(l>>5):
//     one query per lane -> ONE threshold per lane; its 16 rows ascend with r, tiles ascend,
//     so a lane sees train rows in ascending order and the strict filter
//     "distance < current k-th distance" is exact under the lower-row tie rule.
//   top-k: 32 sorted keys in VGPRs per lane (one list per (query, row-half)); the two
//     halves and the train segments are merged by knn_merge_kernel.
//
// Block = 8 waves = 256 queries sharing the A tiles through LDS (super-tiles of 128 rows =
// 16 KB, double buffered, one barrier per super-tile).  Fast path per tile and wave:
// 4 ds_read_b128 + 4 MFMA + 8 v_max3 + 1 compare.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knn.hip.h"

namespace slideo {

typedef int knn_v8i __attribute__((ext_vector_type(8)));
typedef float knn_v16f __attribute__((ext_vector_type(16)));

constexpr int KM_WAVES = 8;                    // waves per block
constexpr int KM_THREADS = KM_WAVES * 64;
constexpr int KM_QPB = KM_WAVES * 32;          // queries per block
constexpr int KM_ST_ROWS = 128;                // rows per super-tile
constexpr int KM_ST_U4 = KM_ST_ROWS * 128 / 16;  // uint4 per super-tile (1024)

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
// Grid (ceil(nq / 256), nseg), block 512.  Segment s covers super-tiles [s*st_per_seg, ...).
// out: [(seg*2 + half)][nq][32] keys.
__global__ __launch_bounds__(KM_THREADS, 2) void knn_mfma_kernel(const uint32_t* __restrict__ q, int nq,
                                                                 const uint4* __restrict__ tx, int nt, int nt_pad,
                                                                 int st_per_seg, uint32_t* __restrict__ out) {
    __shared__ uint4 lds[2][KM_ST_U4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5;
    const int qi = blockIdx.x * KM_QPB + wave * 32 + (lane & 31);
    const int seg = blockIdx.y;
    const int n_st = nt_pad / KM_ST_ROWS;
    const int st0 = seg * st_per_seg, st1 = min(n_st, st0 + st_per_seg);

    // B operand: this lane's query, dwords 2s + half expanded to 32 nibbles each
    knn_v8i bq[4];
    {
        const uint32_t* qp = q + (size_t)min(qi, nq - 1) * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint32_t w = qp[2 * s + half];
            bq[s] = knn_v8i{(int)fp4_expand8(w), (int)fp4_expand8(w >> 8), (int)fp4_expand8(w >> 16), (int)fp4_expand8(w >> 24), 0, 0, 0, 0};
        }
    }
    uint32_t lst[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) lst[i] = KNN_EMPTY;
    float thr = -1024.f;     // dot > thr  <=>  distance < current k-th distance

    if (st0 < st1) {
        // prologue: first super-tile -> LDS[0]
        lds[0][tid] = tx[(size_t)st0 * KM_ST_U4 + tid];
        lds[0][tid + KM_THREADS] = tx[(size_t)st0 * KM_ST_U4 + tid + KM_THREADS];
    }
    __syncthreads();
    int cur = 0;
    for (int st = st0; st < st1; ++st) {
        // stage the next super-tile through registers (latency hidden behind the 16 MFMAs below)
        uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;
        const bool more = st + 1 < st1;
        if (more) {
            n0 = tx[(size_t)(st + 1) * KM_ST_U4 + tid];
            n1 = tx[(size_t)(st + 1) * KM_ST_U4 + tid + KM_THREADS];
        }
        const uint4* L = lds[cur];
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            knn_v16f acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                uint4 a = L[tile * 256 + s * 64 + lane];
                knn_v8i av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bq[s], acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            }
            float m0 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
            float m1 = fmaxf(fmaxf(acc[3], acc[4]), acc[5]);
            float m2 = fmaxf(fmaxf(acc[6], acc[7]), acc[8]);
            float m3 = fmaxf(fmaxf(acc[9], acc[10]), acc[11]);
            float m4 = fmaxf(fmaxf(acc[12], acc[13]), acc[14]);
            float mx = fmaxf(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)), fmaxf(m4, acc[15]));
            if (__builtin_amdgcn_ballot_w64(mx > thr) != 0ull) {
                const int row0 = st * KM_ST_ROWS + tile * 32 + 4 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2);
                    const bool better = acc[r] > thr && row < nt;
                    if (__builtin_amdgcn_ballot_w64(better) != 0ull) {
                        const uint32_t h2 = (uint32_t)(256 - (int)acc[r]);          // = 2 * distance
                        knn_insert<32>(lst, better ? ((h2 << (KNN_KEY_SHIFT - 1)) | (uint32_t)row) : KNN_EMPTY);
                        thr = 256.f - 2.f * (float)(lst[31] >> KNN_KEY_SHIFT);
                    }
                }
            }
        }
        if (more) {
            lds[cur ^ 1][tid] = n0;
            lds[cur ^ 1][tid + KM_THREADS] = n1;
        }
        __syncthreads();
        cur ^= 1;
    }
    if (qi < nq) {
        uint4* o = reinterpret_cast<uint4*>(out + ((size_t)(seg * 2 + half) * nq + qi) * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = make_uint4(lst[4 * i], lst[4 * i + 1], lst[4 * i + 2], lst[4 * i + 3]);
    }
}

}  // namespace slideo
