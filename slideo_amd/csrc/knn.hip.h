// knn.hip.h — exact Hamming k-NN over 256-bit ORB descriptors (gfx950).
//
// Replaces FlannMatcher::knn_match (crates/matching-opencv/src/flann.rs:73-89,
// call site lib.rs:266) with the exact brute-force search north_star asks for
// ([OCV A.8] BFMatcher(NORM_HAMMING).knnMatch semantics: ascending distance,
// ties to the lower train row).
//
// Result encoding: one u32 key per neighbour, key = distance << 23 | train_row
// (distance <= 256 needs 9 bits; train rows < 2^23).  Keys are unique and
// totally ordered, so "k smallest keys" IS the (distance, row) order and the
// result does not depend on the order in which train rows are visited — which
// is what lets the train set be split over blocks and merged.
//
// Kernel shape (integer-VALU bound, not HBM bound: 16 lane-ops per pair —
// 8 x v_xor_b32 + 8 x v_bcnt_u32_b32 — against 64 B of operands per pair that
// are reused 64x (query in VGPRs) and wave-wide (train row in SGPRs)):
//   - each lane owns ONE query: 8 VGPRs of descriptor + KLIST VGPRs of sorted keys;
//   - the train row is wave-uniform, fetched with scalar loads (s_load_dwordx8)
//     and applied as an SGPR operand of v_xor — no LDS, no vector memory in the loop;
//   - insertion is KLIST independent v_med3_u32 (one per slot), executed only when
//     some lane's new key beats its current k-th key (rare after warm-up:
//     ~k ln(M/k) per query).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "types.h"

namespace slideo {


// Insert `key` into the ascending list: new[i] = median(old[i-1], old[i], key),
// new[0] = min(old[0], key).  Swept from the top slot down it is in place and
// every slot is ONE v_med3_u32 with no serial dependence between slots.  The
// tied "+v" operand pins each slot to its register (plain C++ min/max made
// hipcc ping-pong the whole list between two register banks every pair).
// Inserting KNN_EMPTY leaves the list unchanged.
template <int KLIST>
__device__ __forceinline__ void knn_insert(uint32_t (&lst)[KLIST], uint32_t key) {
#pragma unroll
    for (int i = KLIST - 1; i >= 1; --i)
        asm volatile("v_med3_u32 %0, %1, %0, %2" : "+v"(lst[i]) : "v"(lst[i - 1]), "v"(key));
    asm volatile("v_min_u32 %0, %0, %1" : "+v"(lst[0]) : "v"(key));
}

// Offer `key` to the lane's list.  The branch is WAVE-uniform (taken when any
// lane improves); lanes that do not improve insert KNN_EMPTY, which the chain
// leaves the list unchanged for.  This keeps the list registers updated in
// place (a divergent `if` made hipcc copy all KLIST registers per pair).
template <int KLIST>
__device__ __forceinline__ void knn_offer(uint32_t (&lst)[KLIST], uint32_t key) {
    const bool better = key < lst[KLIST - 1];
    if (__builtin_amdgcn_ballot_w64(better) != 0ull) knn_insert<KLIST>(lst, better ? key : KNN_EMPTY);
}

// q: [nq][8] u32, t: [nt][8] u32.  Grid: (ceil(nq/256), nseg).  Block 256.
// out: [nseg][nq][KLIST] keys, ascending.  Segment s covers train rows
// [s*seg_len, min(nt,(s+1)*seg_len)).
template <int KLIST>
__global__ __launch_bounds__(KNN_BLOCK) void knn_hamming_kernel(
    const uint32_t* __restrict__ q, int nq, const uint32_t* __restrict__ t, int nt, int seg_len,
    uint32_t* __restrict__ out) {
    const int qi = blockIdx.x * KNN_BLOCK + threadIdx.x;
    const int seg = blockIdx.y;
    uint32_t qv[8];
    {
        const uint4* qp = reinterpret_cast<const uint4*>(q) + (size_t)min(qi, nq - 1) * 2;
        uint4 a = qp[0], b = qp[1];
        qv[0] = a.x; qv[1] = a.y; qv[2] = a.z; qv[3] = a.w;
        qv[4] = b.x; qv[5] = b.y; qv[6] = b.z; qv[7] = b.w;
    }
    uint32_t lst[KLIST];
#pragma unroll
    for (int i = 0; i < KLIST; ++i) lst[i] = KNN_EMPTY;

    const int j0 = seg * seg_len;
    const int j1 = min(nt, j0 + seg_len);
    // wave-uniform pointer -> scalar loads
    const uint32_t* __restrict__ tp = t + (size_t)j0 * 8;
    int j = j0;
    constexpr int U = 4;
    for (; j + U <= j1; j += U, tp += 8 * U) {
        uint32_t d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t acc = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) acc += __popc(qv[w] ^ tp[u * 8 + w]);
            d[u] = (acc << KNN_KEY_SHIFT) | (uint32_t)(j + u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) knn_offer<KLIST>(lst, d[u]);
    }
    for (; j < j1; ++j, tp += 8) {
        uint32_t acc = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) acc += __popc(qv[w] ^ tp[w]);
        uint32_t key = (acc << KNN_KEY_SHIFT) | (uint32_t)j;
        knn_offer<KLIST>(lst, key);
    }
    if (qi < nq) {
        uint32_t* o = out + ((size_t)seg * nq + qi) * KLIST;
#pragma unroll
        for (int i = 0; i < KLIST; ++i) o[i] = lst[i];
    }
}

// Merge nseg sorted lists per query into the first one (in place into seg 0).
// nq_dev != null: the query count lives on the device (grid sized by capacity), `nq` is ignored
template <int KLIST>
__global__ __launch_bounds__(KNN_BLOCK) void knn_merge_kernel(uint32_t* __restrict__ lists, int nq, int nseg, const uint32_t* __restrict__ nq_dev = nullptr) {
    if (nq_dev) nq = (int)*nq_dev;
    const int qi = blockIdx.x * KNN_BLOCK + threadIdx.x;
    if (qi >= nq) return;
    uint32_t lst[KLIST];
    uint32_t* o = lists + (size_t)qi * KLIST;
#pragma unroll
    for (int i = 0; i < KLIST; ++i) lst[i] = o[i];
    for (int s = 1; s < nseg; ++s) {
        const uint32_t* in = lists + ((size_t)s * nq + qi) * KLIST;
        for (int i = 0; i < KLIST; ++i) {
            uint32_t key = in[i];
            if (key >= lst[KLIST - 1]) break;   // input ascending: nothing later can enter
            knn_insert<KLIST>(lst, key);
        }
    }
#pragma unroll
    for (int i = 0; i < KLIST; ++i) o[i] = lst[i];
}

// Train-set de-duplication (exactness preserving).  Slide decks repeat templates, so many of the pages' 256-bit descriptors
// are IDENTICAL rows of the train matrix (21 % of the headline set).  The matcher searches the unique rows only — a key then
// carries the LOWEST original row of the group of equal rows — and this kernel restores what a search over all rows returns:
// every further member of a key's group is inserted with the same distance (grp_next[row] = the next higher row with the same
// descriptor, -1 at the end of a group).  Why the k best unique rows suffice: the members of a group share one distance, so the
// rows of the true top k at the k-th distance are the lowest rows at that distance, their groups are exactly the groups whose
// lowest row is at most the k-th row, and those come first among the groups of that distance — at most k groups in all.
// Lists are [nq][KLIST] ascending keys, in place.  nq_dev != null: the query count lives on the device.
template <int KLIST>
__global__ __launch_bounds__(KNN_BLOCK) void knn_expand_dups_kernel(uint32_t* __restrict__ lists, int nq, const int32_t* __restrict__ grp_next,
                                                                    const uint32_t* __restrict__ nq_dev = nullptr) {
    if (nq_dev) nq = (int)*nq_dev;
    const int qi = blockIdx.x * KNN_BLOCK + threadIdx.x;
    if (qi >= nq) return;
    uint32_t* o = lists + (size_t)qi * KLIST;
    uint32_t lst[KLIST];
    int32_t nx[KLIST];
    {
        const uint4* o4 = reinterpret_cast<const uint4*>(o);
#pragma unroll
        for (int i = 0; i < KLIST / 4; ++i) { const uint4 v = o4[i]; lst[4 * i] = v.x; lst[4 * i + 1] = v.y; lst[4 * i + 2] = v.z; lst[4 * i + 3] = v.w; }
    }
    bool any = false;
#pragma unroll
    for (int i = 0; i < KLIST; ++i) { nx[i] = lst[i] != KNN_EMPTY ? grp_next[lst[i] & KNN_IDX_MASK] : -1; any = any || nx[i] >= 0; }
    if (!any) return;
    uint32_t dkey[KLIST];
#pragma unroll
    for (int i = 0; i < KLIST; ++i) dkey[i] = lst[i] & ~KNN_IDX_MASK;               // the heads' distances (the list changes below)
#pragma unroll
    for (int i = 0; i < KLIST; ++i) {
        int32_t row = nx[i];
        while (row >= 0) {
            const uint32_t key = dkey[i] | (uint32_t)row;
            if (key >= lst[KLIST - 1]) break;                                       // members ascend: nothing later can enter
            knn_insert<KLIST>(lst, key);
            row = grp_next[row];
        }
    }
    uint4* o4 = reinterpret_cast<uint4*>(o);
#pragma unroll
    for (int i = 0; i < KLIST / 4; ++i) o4[i] = make_uint4(lst[4 * i], lst[4 * i + 1], lst[4 * i + 2], lst[4 * i + 3]);
}

// Debug-tap unpack: keys [nq][KLIST] -> idx [nq][k] (i32, -1 pad), dist [nq][k] (u16, 65535 pad)
__global__ void knn_unpack_kernel(const uint32_t* __restrict__ keys, int nq, int klist, int k,
                                  int32_t* __restrict__ idx, uint16_t* __restrict__ dist) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * k) return;
    int qi = i / k, r = i - qi * k;
    uint32_t key = keys[(size_t)qi * klist + r];
    if (key == KNN_EMPTY) { idx[i] = -1; dist[i] = 65535; }
    else { idx[i] = (int32_t)(key & KNN_IDX_MASK); dist[i] = (uint16_t)(key >> KNN_KEY_SHIFT); }
}

}  // namespace slideo
