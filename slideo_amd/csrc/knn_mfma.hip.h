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
//   top-k: 32 sorted keys per query in the OUTPUT buffer; candidates are pushed to per-lane pending buffers in a
//     global workspace and merged by the query's owner lane in a flush (see the kernel comment); train segments
//     are merged by knn_merge_kernel.
//
// Block = 8 waves = 512 queries sharing the A tiles through LDS: super-tiles of 128 rows = 16 KB go through a
// 4-slot ring filled by LDS-DMA and guarded by per-slot counters — no block barrier in the main loop.  Fast path
// per tile and wave: 4 ds_read_b128 + 8 MFMA + 2 x 8 v_max3_i32 + 2 compares; the two accumulators of a wave run half
// a tile apart so that each one's max tree is issued between the other's MFMAs.
#pragma once
#include <limits.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knn.hip.h"

namespace slideo {

typedef int knn_v8i __attribute__((ext_vector_type(8)));
typedef float knn_v16f __attribute__((ext_vector_type(16)));

constexpr int KM_WAVES = 8;                    // waves per block
constexpr int KM_THREADS = KM_WAVES * 64;
constexpr int KM_QPB = KM_WAVES * 64;          // queries per block (two 32-query B tiles per wave)
constexpr int KM_ST_ROWS = 128;                // rows per super-tile (the unit of LDS staging)
constexpr int KM_RING = 4;                     // LDS ring slots (super-tiles resident per block)
constexpr int KM_AHEAD = 2;                    // a super-tile is staged this many iterations before it is consumed
#ifndef KM_SLEEP
#define KM_SLEEP 1
#endif
#ifndef KM_MFMA_PRIO_V
#define KM_MFMA_PRIO_V 1
#endif
constexpr int KM_MFMA_PRIO = KM_MFMA_PRIO_V;   // wave priority while its MFMAs are issued (0 elsewhere)
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
// prune_tol: 0 = exact k-NN lists; > 0 = lists are exact only for the neighbours with d < best * prune_tol (see flush).
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
// Slow path: a lane whose accumulator beats the threshold PUSHES the candidate key into its own pending buffer
// (one per lane and query tile, global memory, count in a register); nothing is inserted yet.  When some lane has
// KM_FLUSH_AT keys pending the wave flushes between two tiles: every owner lane loads its list, inserts the keys
// its two source lanes pushed (v_med3 chain, KM_FLUSH_BATCH loads in flight), stores the list, and the new
// thresholds are re-broadcast.  This batches the divergent part: an insert costs the whole wave ~35 VALU
// instructions, and without batching it ran for what is usually ONE lane's candidate.
// Exactness: thresholds only tighten — in a flush, which runs between tiles, and (prune_tol > 0) right after a slow
// path from the best row just pushed; every row of the tile being
// scored is higher than every row already in a list, so a candidate with exactly the k-th distance is (correctly)
// rejected by the strict filter, and nothing that belongs to the final top-k is ever filtered out.  With
// prune_tol > 0 the threshold is additionally capped by the vote's acceptance bound (see the flush).
constexpr int KM_FLUSH_AT = 16;
constexpr int KM_FLUSH_BATCH = 4;
constexpr int KM_PEND_CAP = 64;                      // >= KM_FLUSH_AT - 1 + 2 * 16 (flush test every second tile; a lane pushes <= 16 keys per tile and query)
constexpr size_t KM_PEND_WORDS_PER_WAVE = (size_t)2 * KM_PEND_CAP * 64;

// Pending buffers: every lane has a PRIVATE buffer per query tile in a global workspace (L2 resident; the
// count lives in a register, no atomics).  The owner of tile-0 query n (lane n) gathers from lanes n and
// n+32, the owner of tile-1 query n (lane 32+n) likewise.  Layout [wave][tile][slot][lane].
// The sorted list of a query lives in the output buffer and is only brought into registers inside a flush,
// which runs between tiles (accumulators dead) when some lane has >= KM_FLUSH_AT keys pending.
__global__ __launch_bounds__(KM_THREADS, 4) void knn_mfma_kernel(const uint32_t* __restrict__ q, int nq,
                                                                 const uint4* __restrict__ tx, int nt, int nt_pad,
                                                                 int st_per_seg, uint32_t* __restrict__ out,
                                                                 uint32_t* __restrict__ pend_ws, float prune_tol
#ifdef KM_TIMING
                                                                 , unsigned long long* dbg
#endif
                                                                 ) {
#ifdef KM_TIMING
    unsigned long long t_bar = 0, t_flush = 0, t_slow = 0, n_flush = 0, n_slow = 0;
    const unsigned long long t_begin = __builtin_readcyclecounter();
#define KM_T0 const unsigned long long t0_ = __builtin_readcyclecounter();
#define KM_T1(acc) acc += __builtin_readcyclecounter() - t0_;
#else
#define KM_T0
#define KM_T1(acc)
#endif
    __shared__ uint4 lds[KM_RING][KM_ST_U4];
    __shared__ uint32_t s_filled[KM_RING], s_done[KM_RING];   // waves that wrote / finished reading each slot (monotonic)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: SGPR bases below)
    const int half = lane >> 5, ql = lane & 31;
    const int qbase = blockIdx.x * KM_QPB + wave * 64;
    const int qi = qbase + lane;                                    // the query whose list this lane owns
    const int seg = blockIdx.y;
    const int n_st = nt_pad / KM_ST_ROWS;
    const int st0 = seg * st_per_seg, st1 = min(n_st, st0 + st_per_seg);
    uint32_t* const PA = pend_ws + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * KM_WAVES + wave) * KM_PEND_WORDS_PER_WAVE;
    uint32_t* const PB = PA + (size_t)KM_PEND_CAP * 64;

    knn_v8i bq0[4], bq1[4];
    auto load_queries = [&]() {
        const uint32_t* q0 = q + (size_t)min(qbase + ql, nq - 1) * 8;
        const uint32_t* q1 = q + (size_t)min(qbase + 32 + ql, nq - 1) * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint32_t w0 = q0[2 * s + half], w1 = q1[2 * s + half];
            bq0[s] = knn_v8i{(int)fp4_expand8(w0), (int)fp4_expand8(w0 >> 8), (int)fp4_expand8(w0 >> 16), (int)fp4_expand8(w0 >> 24), 0, 0, 0, 0};
            bq1[s] = knn_v8i{(int)fp4_expand8(w1), (int)fp4_expand8(w1 >> 8), (int)fp4_expand8(w1 >> 16), (int)fp4_expand8(w1 >> 24), 0, 0, 0, 0};
        }
    };
    load_queries();
    uint4* const my_list = reinterpret_cast<uint4*>(out + ((size_t)seg * nq + min(qi, nq - 1)) * 32);
    const bool owner_valid = qi < nq;
    if (owner_valid) {
#pragma unroll
        for (int i = 0; i < 8; ++i) my_list[i] = make_uint4(KNN_EMPTY, KNN_EMPTY, KNN_EMPTY, KNN_EMPTY);
    }
    float thrA = -1024.f, thrB = -1024.f;     // dot > thr  <=>  distance < current k-th distance of that query
    int thrAi = INT_MIN, thrBi = INT_MIN;     // the same for the bit-pattern compare of the fast path (INT_MIN while thr < 0)
    uint32_t cntA = 0, cntB = 0;              // keys pending in this lane's private buffers

    // owners drain the pending buffers of their two source lanes into their sorted list
    auto flush = [&]() {
        const uint32_t cA_lo = __shfl(cntA, ql), cA_hi = __shfl(cntA, ql + 32);
        const uint32_t cB_lo = __shfl(cntB, ql), cB_hi = __shfl(cntB, ql + 32);
        // the sibling waves of the block can run at most KM_AHEAD super-tiles ahead of a flushing wave, so the flush
        // is the block's critical path: give it the SIMD's issue slots
        const uint32_t c_lo = half ? cB_lo : cA_lo, c_hi = half ? cB_hi : cA_hi;
        const uint32_t* PP = (half ? PB : PA) + ql;
        uint32_t lst[32];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 v = my_list[i];
            lst[4 * i] = v.x; lst[4 * i + 1] = v.y; lst[4 * i + 2] = v.z; lst[4 * i + 3] = v.w;
        }
        // pending keys are fetched KM_FLUSH_BATCH at a time so that their (L2) latency is paid once per batch, not
        // once per key; a key slot past a lane's count reads as KNN_EMPTY, whose insertion is a no-op
        for (uint32_t base = 0; __builtin_amdgcn_ballot_w64(base < c_lo) != 0ull; base += KM_FLUSH_BATCH) {
            uint32_t e[KM_FLUSH_BATCH];
#pragma unroll
            for (int i = 0; i < KM_FLUSH_BATCH; ++i) e[i] = base + i < c_lo ? PP[(base + i) * 64] : KNN_EMPTY;
#pragma unroll
            for (int i = 0; i < KM_FLUSH_BATCH; ++i) knn_insert<32>(lst, e[i]);
        }
        for (uint32_t base = 0; __builtin_amdgcn_ballot_w64(base < c_hi) != 0ull; base += KM_FLUSH_BATCH) {
            uint32_t e[KM_FLUSH_BATCH];
#pragma unroll
            for (int i = 0; i < KM_FLUSH_BATCH; ++i) e[i] = base + i < c_hi ? PP[(base + i) * 64 + 32] : KNN_EMPTY;
#pragma unroll
            for (int i = 0; i < KM_FLUSH_BATCH; ++i) knn_insert<32>(lst, e[i]);
        }
        if (owner_valid) {
#pragma unroll
            for (int i = 0; i < 8; ++i) my_list[i] = make_uint4(lst[4 * i], lst[4 * i + 1], lst[4 * i + 2], lst[4 * i + 3]);
        }
        float t = 256.f - 2.f * (float)(lst[31] >> KNN_KEY_SHIFT);
        if (prune_tol > 0.f) {
            // Fused vote filter: the only consumer of the lists keeps a neighbour iff (float)d < (float)best * tol
            // (f32, strict; best = the query's smallest distance).  best only decreases, so a row that fails the test
            // against the CURRENT best can never pass it: it is not worth a list slot.  d < lim  <=>  d <= ceil(lim) - 1.
            const float lim = (float)(lst[0] >> KNN_KEY_SHIFT) * prune_tol;       // (empty list: 511 * tol, no filter)
            t = fmaxf(t, 255.f - 2.f * (ceilf(lim) - 1.f));
        }
        cntA = 0; cntB = 0;
        thrA = __shfl(t, ql);
        thrB = __shfl(t, 32 + ql);
        thrAi = thrA >= 0.f ? __float_as_int(thrA) : INT_MIN;
        thrBi = thrB >= 0.f ? __float_as_int(thrB) : INT_MIN;
    };

    // No block barrier in the main loop: a barrier per super-tile kept the 8 waves in lock-step and cost a
    // third of every wave's lifetime (one wave in the slow path or a flush stalls the other seven).  Instead the
    // super-tiles go through a KM_RING-slot LDS ring guarded by two monotonic counters per slot:
    //   s_filled[slot] += 1 by each wave once its share of a super-tile is written (consume when == 8 * use#)
    //   s_done[slot]   += 1 by each wave once it has finished reading the slot       (overwrite when == 8 * use#)
    // A wave starts the DMA of its share of super-tile j + KM_AHEAD at the top of its iteration j and publishes it
    // at the end of that iteration (s_waitcnt vmcnt(0), then the count), so waves may drift apart by about
    // KM_AHEAD super-tiles in either direction before anyone waits.
    // this wave's share (2 x 1 KiB) of a super-tile: global -> LDS by DMA (no staging registers, no ds_write pass);
    // lane i of a wave-instruction lands at the wave-uniform LDS base + 16 i
    auto stage = [&](int jj, int sl) {
        constexpr int PER_WAVE = KM_ST_U4 / KM_WAVES;                  // uint4 per wave and super-tile
        static_assert(PER_WAVE % 64 == 0, "whole wave-instructions");
        const uint4* src = tx + (size_t)(st0 + jj) * KM_ST_U4 + wave * PER_WAVE + lane;
#pragma unroll
        for (int i = 0; i < PER_WAVE / 64; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * i),
                                             (__attribute__((address_space(3))) void*)&lds[sl][wave * PER_WAVE + 64 * i], 16, 0, 0);
    };
    auto signal = [&](uint32_t* f) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wait_ge = [&](uint32_t* f, uint32_t target) {
        KM_T0
        while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
            __builtin_amdgcn_s_sleep(KM_SLEEP);
        asm volatile("" ::: "memory");
        KM_T1(t_bar)
    };
    if (tid < KM_RING) { s_filled[tid] = 0; s_done[tid] = 0; }
    __syncthreads();
    const int nst = st1 - st0;
    for (int j = 0; j < KM_AHEAD && j < nst; ++j) stage(j, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int j = 0; j < KM_AHEAD && j < nst; ++j) signal(&s_filled[j]);
    // Skewed main loop.  The two accumulators of a wave run half a tile apart: while the four MFMAs of one are in the
    // matrix pipe the wave's own VALU slots take the other's epilogue (max tree + threshold test), interleaved in
    // program order (sched_group_barrier), so a wave keeps the pipe busy by itself instead of relying on a sibling
    // wave being in its epilogue at the right moment (with two blocks per CU the younger block ran alone, two waves
    // per SIMD, for the last quarter of the kernel with the pipe under half busy).  Tile t of the segment:
    //     half 1:  a1 = F(t) x tile-1 queries      ||  test a0 (tile t)       -> candidates of a0
    //     half 2:  a0 = F(t+1) x tile-0 queries    ||  test a1 (tile t)       -> candidates of a1
    // a0 therefore reads the ring one tile ahead of a1: a slot is acquired when a0 enters its super-tile and released
    // when a1 leaves it.  Rows past the end of the train set are not masked in the test (a spurious trip to the slow
    // path in the last tile at worst); the slow path drops them by row index.
    constexpr int TPS = KM_ST_ROWS / 32;                               // tiles per super-tile
    const int T = nst * TPS;
    auto acquire = [&](int j) {                                        // a0 is about to read super-tile j
        const int jp = j - 1 + KM_AHEAD, jn = j + KM_AHEAD;
        if (j > 0 && jp < nst) {                                       // publish this wave's share staged at acquire(j - 1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            signal(&s_filled[jp % KM_RING]);
        }
        if (jn < nst) {                                                // slot was last read for super-tile jn - KM_RING
            wait_ge(&s_done[jn % KM_RING], (uint32_t)KM_WAVES * (uint32_t)(jn / KM_RING));
            stage(jn, jn % KM_RING);
        }
        wait_ge(&s_filled[j % KM_RING], (uint32_t)KM_WAVES * (uint32_t)(j / KM_RING + 1));
    };
    auto frag = [&](int t, int s) -> uint4 { return lds[(t / TPS) % KM_RING][(t % TPS) * 256 + 64 * s + lane]; };
    auto mfma = [&](knn_v16f acc, const uint4& f, const knn_v8i& b) {
        const knn_v8i v = {(int)f.x, (int)f.y, (int)f.z, (int)f.w, 0, 0, 0, 0};
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v, b, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    };
    // maxima of the raw bit patterns (see the fast-path note below): triples {3k, 3k+1, 3k+2}, k = 0..4, register 15 apart
    auto tree = [&](const knn_v16f& acc, int* tk) -> int {
#pragma unroll
        for (int k = 0; k < 5; ++k)
            tk[k] = max(max(__float_as_int(acc[3 * k]), __float_as_int(acc[3 * k + 1])), __float_as_int(acc[3 * k + 2]));
        return max(max(max(__float_as_int(acc[15]), tk[0]), tk[1]), max(max(tk[2], tk[3]), tk[4]));
    };
    // Fast path test: for thr >= 0 the integer order of the bit patterns decides "v > thr" exactly (a negative v has the
    // sign bit set and compares below every thr >= 0; non-negative floats order like their bits); a threshold < 0 (list
    // not full yet, or tiny train sets) is kept as INT_MIN so that everything goes to the exact slow path.
    // Slow path (about one tile in twenty): usually ONE value of ONE lane qualifies, so each of the candidate tests is
    // a wave-uniform "nobody" branch that falls through.
    auto candidates = [&](const knn_v16f& acc, const int* tk, int t, float& thr, int& thri, uint32_t* P, uint32_t& cnt) {
        KM_T0
#ifdef KM_TIMING
        ++n_slow;
#endif
        const uint32_t row0 = (uint32_t)((st0 * TPS + t) * 32 + 4 * half);
        float vbest = -1024.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const bool gate = k < 5 ? tk[k < 5 ? k : 0] > thri : __float_as_int(acc[15]) > thri;
            if (__builtin_amdgcn_ballot_w64(gate) == 0ull) continue;
#pragma unroll
            for (int r = 3 * k; r < 3 * k + 3 && r < 16; ++r) {
                const uint32_t row = row0 + (r & 3) + 8 * (r >> 2);
                const float v = acc[r];
                const bool h = v > thr && row < (uint32_t)nt;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(h) != 0ull, 0)) {
                    if (h) {
                        P[cnt * 64 + lane] = ((uint32_t)(256 - (int)v) << (KNN_KEY_SHIFT - 1)) | row;
                        ++cnt;
                        vbest = fmaxf(vbest, v);
                    }
                }
            }
        }
        if (prune_tol > 0.f) {
            // Fused vote filter, applied at once instead of at the next flush: every distance seen so far bounds the
            // query's final best from above, so the acceptance bound of the best row this lane just pushed is already
            // valid, and so is the one its partner lane (the other 16 rows of the same query) derived.  A looser
            // threshold is never wrong, so nothing else has to agree.
            float tn = 255.f - 2.f * (ceilf((256.f - vbest) * 0.5f * prune_tol) - 1.f);
            tn = fmaxf(tn, __shfl_xor(tn, 32));
            thr = fmaxf(thr, tn);
            thri = thr >= 0.f ? __float_as_int(thr) : INT_MIN;
        }
        KM_T1(t_slow)
    };
    if (T > 0) {
        knn_v16f a0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0;
        const knn_v16f zero = a0;
        acquire(0);
        uint4 x0 = frag(0, 0), x1 = frag(0, 1), x2 = frag(0, 2), x3 = frag(0, 3);          // F(t)
        uint4 y0, y1, y2, y3;                                                              // F(t + 1)
        a0 = mfma(zero, x0, bq0[0]); a0 = mfma(a0, x1, bq0[1]); a0 = mfma(a0, x2, bq0[2]); a0 = mfma(a0, x3, bq0[3]);
        // one tile: (c*) = F(t) are live on entry, (n*) = F(t+1) on exit
#define KM_TILE(t, c0, c1, c2, c3, n0, n1, n2, n3)                                                                    \
        {                                                                                                             \
            const int tn_ = min((t) + 1, T - 1);                 /* the tile after the last: recomputed, never tested */ \
            if (tn_ % TPS == 0 && (t) + 1 < T) acquire(tn_ / TPS);                                                    \
            n0 = frag(tn_, 0); n1 = frag(tn_, 1);                                                                     \
            int tk_[5];                                                                                               \
            __builtin_amdgcn_s_setprio(KM_MFMA_PRIO);                                                                 \
            a1 = mfma(zero, c0, bq1[0]); a1 = mfma(a1, c1, bq1[1]); a1 = mfma(a1, c2, bq1[2]); a1 = mfma(a1, c3, bq1[3]); \
            const int m0_ = tree(a0, tk_);                                                                            \
            KM_INTERLEAVE                                                                                             \
            asm volatile("" : "+v"(a1));                         /* keeps the MFMAs above the branch below */          \
            __builtin_amdgcn_s_setprio(0);                                                                            \
            n2 = frag(tn_, 2); n3 = frag(tn_, 3);                                                                     \
            if (__builtin_amdgcn_ballot_w64(m0_ > thrAi) != 0ull) candidates(a0, tk_, (t), thrA, thrAi, PA, cntA);    \
            __builtin_amdgcn_s_setprio(KM_MFMA_PRIO);                                                                 \
            a0 = mfma(zero, n0, bq0[0]); a0 = mfma(a0, n1, bq0[1]); a0 = mfma(a0, n2, bq0[2]); a0 = mfma(a0, n3, bq0[3]); \
            const int m1_ = tree(a1, tk_);                                                                            \
            KM_INTERLEAVE                                                                                             \
            asm volatile("" : "+v"(a0));                                                                              \
            __builtin_amdgcn_s_setprio(0);                                                                            \
            if (__builtin_amdgcn_ballot_w64(m1_ > thrBi) != 0ull) candidates(a1, tk_, (t), thrB, thrBi, PB, cntB);    \
            if (((t) + 1) % TPS == 0) signal(&s_done[((t) / TPS) % KM_RING]);                                         \
        }
        // one MFMA, then two of the other accumulator's max3, four times over
#define KM_INTERLEAVE                                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
#pragma unroll 1
        for (int t = 0; t < T; t += 2) {                         // T is a multiple of TPS = 4
            KM_TILE(t, x0, x1, x2, x3, y0, y1, y2, y3)
            KM_TILE(t + 1, y0, y1, y2, y3, x0, x1, x2, x3)
            if (__builtin_amdgcn_ballot_w64(cntA >= (uint32_t)KM_FLUSH_AT || cntB >= (uint32_t)KM_FLUSH_AT) != 0ull) {
                KM_T0
                flush();
                KM_T1(t_flush)
#ifdef KM_TIMING
                ++n_flush;
#endif
                // everything the loop carries is rebuilt after the (rare) flush instead of kept alive across it: the flush
                // needs 32 registers for the list, and values live across it would be spilled on every path
                load_queries();
                const int tr = min(t + 2, T - 1);
                x0 = frag(tr, 0); x1 = frag(tr, 1); x2 = frag(tr, 2); x3 = frag(tr, 3);
                a0 = mfma(zero, x0, bq0[0]); a0 = mfma(a0, x1, bq0[1]); a0 = mfma(a0, x2, bq0[2]); a0 = mfma(a0, x3, bq0[3]);
            }
        }
#undef KM_TILE
#undef KM_INTERLEAVE
    }
    flush();
#ifdef KM_TIMING
    if (lane == 0) {
        atomicAdd(dbg + 0, __builtin_readcyclecounter() - t_begin);
        atomicAdd(dbg + 1, t_bar); atomicAdd(dbg + 2, t_flush); atomicAdd(dbg + 3, t_slow);
        atomicAdd(dbg + 4, n_flush); atomicAdd(dbg + 5, n_slow); atomicAdd(dbg + 6, 1ull);
    }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------
// knn_mfma4_kernel — the same algorithm with FOUR query tiles per wave and two waves per SIMD.
//
// A resident knn_mfma_kernel holds 4 waves x 115 VGPRs of every SIMD lane and 131 KB of the CU's LDS, so the kernels of
// the other batch in flight (ORB, verify) mostly run in its tail.  Here a wave owns 128 queries (4 B tiles, 4
// accumulators), a block of 8 waves 1024 queries, one block per CU: per SIMD two waves x ~190 VGPRs, per CU 64 KB of
// LDS — room for two to three foreign waves per SIMD and 96 KB of LDS throughout the launch — and every A fragment read
// from LDS feeds four MFMAs instead of two.  With only two waves per SIMD nobody else hides a wave's epilogue, so the
// accumulators run skewed in two groups: while the 8 MFMAs of {a2, a3} are in the pipe the wave's VALU slots take the
// max trees and threshold tests of {a0, a1}, and vice versa (see knn_mfma_kernel for the skew, the ring, the slow path
// and the flush, which are the same).  Lane L owns the lists of wave-local queries L (tile L/32) and L + 64 (tile 2 + L/32).
// Grid (ceil(nq / 1024), nseg), block 512.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef K4_PRIO
#define K4_PRIO KM_MFMA_PRIO
#endif
constexpr int K4_NT = 4;
constexpr int K4_QPW = 32 * K4_NT;                       // queries per wave
constexpr int K4_QPB = KM_WAVES * K4_QPW;                // queries per block
constexpr size_t K4_PEND_WORDS_PER_WAVE = (size_t)K4_NT * KM_PEND_CAP * 64;

__global__ __launch_bounds__(KM_THREADS, 2) void knn_mfma4_kernel(const uint32_t* __restrict__ q, int nq,
                                                                  const uint4* __restrict__ tx, int nt, int nt_pad,
                                                                  int st_per_seg, uint32_t* __restrict__ out,
                                                                  uint32_t* __restrict__ pend_ws, float prune_tol) {
    __shared__ uint4 lds[KM_RING][KM_ST_U4];
    __shared__ uint32_t s_filled[KM_RING], s_done[KM_RING];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, ql = lane & 31;
    const int qbase = blockIdx.x * K4_QPB + wave * K4_QPW;
    const int seg = blockIdx.y;
    const int n_st = nt_pad / KM_ST_ROWS;
    const int st0 = seg * st_per_seg, st1 = min(n_st, st0 + st_per_seg);
    uint32_t* const P0 = pend_ws + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * KM_WAVES + wave) * K4_PEND_WORDS_PER_WAVE;
    auto pend = [&](int i) -> uint32_t* { return P0 + (size_t)i * KM_PEND_CAP * 64; };

    knn_v8i bq[K4_NT][4];
    auto load_queries = [&]() {
#pragma unroll
        for (int i = 0; i < K4_NT; ++i) {
            const uint32_t* qp = q + (size_t)min(qbase + 32 * i + ql, nq - 1) * 8;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint32_t w = qp[2 * s + half];
                bq[i][s] = knn_v8i{(int)fp4_expand8(w), (int)fp4_expand8(w >> 8), (int)fp4_expand8(w >> 16), (int)fp4_expand8(w >> 24), 0, 0, 0, 0};
            }
        }
    };
    load_queries();
    // list p of this lane: wave-local query 64 p + lane, i.e. tile 2 p + half, column ql
    auto list_of = [&](int p) -> uint4* { return reinterpret_cast<uint4*>(out + ((size_t)seg * nq + min(qbase + 64 * p + lane, nq - 1)) * 32); };
#pragma unroll
    for (int p = 0; p < K4_NT / 2; ++p)
        if (qbase + 64 * p + lane < nq) {
            uint4* l = list_of(p);
#pragma unroll
            for (int i = 0; i < 8; ++i) l[i] = make_uint4(KNN_EMPTY, KNN_EMPTY, KNN_EMPTY, KNN_EMPTY);
        }
    float thr[K4_NT]; int thri[K4_NT]; uint32_t cnt[K4_NT];
#pragma unroll
    for (int i = 0; i < K4_NT; ++i) { thr[i] = -1024.f; thri[i] = INT_MIN; cnt[i] = 0; }

    auto flush = [&]() {
#pragma unroll
        for (int p = 0; p < K4_NT / 2; ++p) {
            const int A = 2 * p, B = 2 * p + 1;
            const uint32_t cA_lo = __shfl(cnt[A], ql), cA_hi = __shfl(cnt[A], ql + 32);
            const uint32_t cB_lo = __shfl(cnt[B], ql), cB_hi = __shfl(cnt[B], ql + 32);
            const uint32_t c_lo = half ? cB_lo : cA_lo, c_hi = half ? cB_hi : cA_hi;
            const uint32_t* PP = (half ? pend(B) : pend(A)) + ql;
            uint4* my_list = list_of(p);
            const bool owner_valid = qbase + 64 * p + lane < nq;
            uint32_t lst[32];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint4 v = my_list[i];
                lst[4 * i] = v.x; lst[4 * i + 1] = v.y; lst[4 * i + 2] = v.z; lst[4 * i + 3] = v.w;
            }
            for (uint32_t base = 0; __builtin_amdgcn_ballot_w64(base < c_lo) != 0ull; base += KM_FLUSH_BATCH) {
                uint32_t e[KM_FLUSH_BATCH];
#pragma unroll
                for (int i = 0; i < KM_FLUSH_BATCH; ++i) e[i] = base + i < c_lo ? PP[(base + i) * 64] : KNN_EMPTY;
#pragma unroll
                for (int i = 0; i < KM_FLUSH_BATCH; ++i) knn_insert<32>(lst, e[i]);
            }
            for (uint32_t base = 0; __builtin_amdgcn_ballot_w64(base < c_hi) != 0ull; base += KM_FLUSH_BATCH) {
                uint32_t e[KM_FLUSH_BATCH];
#pragma unroll
                for (int i = 0; i < KM_FLUSH_BATCH; ++i) e[i] = base + i < c_hi ? PP[(base + i) * 64 + 32] : KNN_EMPTY;
#pragma unroll
                for (int i = 0; i < KM_FLUSH_BATCH; ++i) knn_insert<32>(lst, e[i]);
            }
            if (owner_valid) {
#pragma unroll
                for (int i = 0; i < 8; ++i) my_list[i] = make_uint4(lst[4 * i], lst[4 * i + 1], lst[4 * i + 2], lst[4 * i + 3]);
            }
            float t = 256.f - 2.f * (float)(lst[31] >> KNN_KEY_SHIFT);
            if (prune_tol > 0.f) {
                const float lim = (float)(lst[0] >> KNN_KEY_SHIFT) * prune_tol;
                t = fmaxf(t, 255.f - 2.f * (ceilf(lim) - 1.f));
            }
            cnt[A] = 0; cnt[B] = 0;
            thr[A] = __shfl(t, ql);
            thr[B] = __shfl(t, 32 + ql);
            thri[A] = thr[A] >= 0.f ? __float_as_int(thr[A]) : INT_MIN;
            thri[B] = thr[B] >= 0.f ? __float_as_int(thr[B]) : INT_MIN;
        }
    };
    auto stage = [&](int jj, int sl) {
        constexpr int PER_WAVE = KM_ST_U4 / KM_WAVES;
        const uint4* src = tx + (size_t)(st0 + jj) * KM_ST_U4 + wave * PER_WAVE + lane;
#pragma unroll
        for (int i = 0; i < PER_WAVE / 64; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * i),
                                             (__attribute__((address_space(3))) void*)&lds[sl][wave * PER_WAVE + 64 * i], 16, 0, 0);
    };
    auto signal = [&](uint32_t* f) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wait_ge = [&](uint32_t* f, uint32_t target) {
        while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
            __builtin_amdgcn_s_sleep(KM_SLEEP);
        asm volatile("" ::: "memory");
    };
    if (tid < KM_RING) { s_filled[tid] = 0; s_done[tid] = 0; }
    __syncthreads();
    const int nst = st1 - st0;
    for (int j = 0; j < KM_AHEAD && j < nst; ++j) stage(j, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int j = 0; j < KM_AHEAD && j < nst; ++j) signal(&s_filled[j]);
    constexpr int TPS = KM_ST_ROWS / 32;
    const int T = nst * TPS;
    auto acquire = [&](int j) {
        const int jp = j - 1 + KM_AHEAD, jn = j + KM_AHEAD;
        if (j > 0 && jp < nst) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            signal(&s_filled[jp % KM_RING]);
        }
        if (jn < nst) {
            wait_ge(&s_done[jn % KM_RING], (uint32_t)KM_WAVES * (uint32_t)(jn / KM_RING));
            stage(jn, jn % KM_RING);
        }
        wait_ge(&s_filled[j % KM_RING], (uint32_t)KM_WAVES * (uint32_t)(j / KM_RING + 1));
    };
    auto frag = [&](int t, int s) -> uint4 { return lds[(t / TPS) % KM_RING][(t % TPS) * 256 + 64 * s + lane]; };
    auto mfma = [&](knn_v16f acc, const uint4& f, const knn_v8i& b) {
        const knn_v8i v = {(int)f.x, (int)f.y, (int)f.z, (int)f.w, 0, 0, 0, 0};
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v, b, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    };
    auto tree = [&](const knn_v16f& acc, int* tk) -> int {
#pragma unroll
        for (int k = 0; k < 5; ++k)
            tk[k] = max(max(__float_as_int(acc[3 * k]), __float_as_int(acc[3 * k + 1])), __float_as_int(acc[3 * k + 2]));
        return max(max(max(__float_as_int(acc[15]), tk[0]), tk[1]), max(max(tk[2], tk[3]), tk[4]));
    };
    auto candidates = [&](const knn_v16f& acc, const int* tk, int t, float& th, int& thi, uint32_t* P, uint32_t& c) {
        const uint32_t row0 = (uint32_t)((st0 * TPS + t) * 32 + 4 * half);
        float vbest = -1024.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const bool gate = k < 5 ? tk[k < 5 ? k : 0] > thi : __float_as_int(acc[15]) > thi;
            if (__builtin_amdgcn_ballot_w64(gate) == 0ull) continue;
#pragma unroll
            for (int r = 3 * k; r < 3 * k + 3 && r < 16; ++r) {
                const uint32_t row = row0 + (r & 3) + 8 * (r >> 2);
                const float v = acc[r];
                const bool h = v > th && row < (uint32_t)nt;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(h) != 0ull, 0)) {
                    if (h) {
                        P[c * 64 + lane] = ((uint32_t)(256 - (int)v) << (KNN_KEY_SHIFT - 1)) | row;
                        ++c;
                        vbest = fmaxf(vbest, v);
                    }
                }
            }
        }
        if (prune_tol > 0.f) {
            float tn = 255.f - 2.f * (ceilf((256.f - vbest) * 0.5f * prune_tol) - 1.f);
            tn = fmaxf(tn, __shfl_xor(tn, 32));
            th = fmaxf(th, tn);
            thi = th >= 0.f ? __float_as_int(th) : INT_MIN;
        }
    };
    if (T > 0) {
        const knn_v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        knn_v16f a[K4_NT];
#pragma unroll
        for (int i = 0; i < K4_NT; ++i) a[i] = zero;
        acquire(0);
        uint4 x0 = frag(0, 0), x1 = frag(0, 1), x2 = frag(0, 2), x3 = frag(0, 3);
        uint4 y0, y1, y2, y3;
        // two accumulator chains through the four k-steps of one A tile: 8 MFMAs
#define K4_MFMA8(G, f0, f1, f2, f3)                                                                                     \
        a[G] = mfma(zero, f0, bq[G][0]); a[G + 1] = mfma(zero, f0, bq[G + 1][0]);                                      \
        a[G] = mfma(a[G], f1, bq[G][1]); a[G + 1] = mfma(a[G + 1], f1, bq[G + 1][1]);                                  \
        a[G] = mfma(a[G], f2, bq[G][2]); a[G + 1] = mfma(a[G + 1], f2, bq[G + 1][2]);                                  \
        a[G] = mfma(a[G], f3, bq[G][3]); a[G + 1] = mfma(a[G + 1], f3, bq[G + 1][3]);
        // one MFMA, then two instructions of the other group's max trees, eight times over
#define K4_INTERLEAVE                                                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        // tests of group G (accumulators G, G+1) for tile t
#define K4_TEST(G, t)                                                                                                   \
        if (__builtin_amdgcn_ballot_w64(mA_ > thri[G] || mB_ > thri[G + 1]) != 0ull) {                                  \
            if (__builtin_amdgcn_ballot_w64(mA_ > thri[G]) != 0ull) candidates(a[G], tkA_, (t), thr[G], thri[G], pend(G), cnt[G]); \
            if (__builtin_amdgcn_ballot_w64(mB_ > thri[G + 1]) != 0ull) candidates(a[G + 1], tkB_, (t), thr[G + 1], thri[G + 1], pend(G + 1), cnt[G + 1]); \
        }
        // one tile: (c*) = F(t) live on entry, (n*) = F(t+1) on exit; group {0,1} already holds tile t
#define K4_TILE(t, c0, c1, c2, c3, n0, n1, n2, n3)                                                                      \
        {                                                                                                             \
            const int tn_ = min((t) + 1, T - 1);                                                                      \
            if (tn_ % TPS == 0 && (t) + 1 < T) acquire(tn_ / TPS);                                                    \
            n0 = frag(tn_, 0); n1 = frag(tn_, 1);                                                                     \
            int tkA_[5], tkB_[5];                                                                                     \
            __builtin_amdgcn_s_setprio(K4_PRIO);                                                                 \
            K4_MFMA8(2, c0, c1, c2, c3)                                                                               \
            int mA_ = tree(a[0], tkA_), mB_ = tree(a[1], tkB_);                                                       \
            K4_INTERLEAVE                                                                                             \
            asm volatile("" : "+v"(a[2]), "+v"(a[3]));                                                                \
            __builtin_amdgcn_s_setprio(0);                                                                            \
            n2 = frag(tn_, 2); n3 = frag(tn_, 3);                                                                     \
            K4_TEST(0, t)                                                                                             \
            __builtin_amdgcn_s_setprio(K4_PRIO);                                                                 \
            K4_MFMA8(0, n0, n1, n2, n3)                                                                               \
            mA_ = tree(a[2], tkA_); mB_ = tree(a[3], tkB_);                                                           \
            K4_INTERLEAVE                                                                                             \
            asm volatile("" : "+v"(a[0]), "+v"(a[1]));                                                                \
            __builtin_amdgcn_s_setprio(0);                                                                            \
            K4_TEST(2, t)                                                                                             \
            if (((t) + 1) % TPS == 0) signal(&s_done[((t) / TPS) % KM_RING]);                                         \
        }
        K4_MFMA8(0, x0, x1, x2, x3)
#pragma unroll 1
        for (int t = 0; t < T; t += 2) {
            K4_TILE(t, x0, x1, x2, x3, y0, y1, y2, y3)
            K4_TILE(t + 1, y0, y1, y2, y3, x0, x1, x2, x3)
            bool need = false;
#pragma unroll
            for (int i = 0; i < K4_NT; ++i) need |= cnt[i] >= (uint32_t)KM_FLUSH_AT;
            if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                flush();
                load_queries();                                  // (rebuilt, not kept alive across the flush: see knn_mfma_kernel)
                const int tr = min(t + 2, T - 1);
                x0 = frag(tr, 0); x1 = frag(tr, 1); x2 = frag(tr, 2); x3 = frag(tr, 3);
                K4_MFMA8(0, x0, x1, x2, x3)
            }
        }
#undef K4_TILE
#undef K4_TEST
#undef K4_INTERLEAVE
#undef K4_MFMA8
    }
    flush();
}

}  // namespace slideo
