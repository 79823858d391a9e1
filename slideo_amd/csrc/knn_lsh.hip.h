// knn_lsh.hip.h — slideo_config.matcher 1: the LSH-compatible approximate search (third item).
//
// The reference does not search exhaustively: FlannMatcher builds FLANN's LshIndex (crates/matching-opencv/src/flann.rs:14-26:
// 6 tables, 12-bit keys, multi-probe level 1) and knn_match (flann.rs:73-89) only ever scores the rows that share — up to one
// flipped key bit — a bucket with the query in some table.  This mode restates that CANDIDATE RULE (flann/lsh_index.h,
// lsh_table.h, recalled; oracle LshIdx is the parity target) and returns the k nearest candidates by (distance, row):
// recall < 1, like the reference's.  Keys and tables are built on the host at finalize (geom.h lsh_params / lsh_key_host).
//
// One wave per query.  A query has ~10^4 candidates (6 tables x 13 probes x M / 4096 rows) of 5 x 10^5 rows; selecting the k
// best WITHOUT sorting them: distances are integers 0..256, so pass A histograms the distances of the (unique) candidates in
// LDS, a wave scan finds the distance D* of the k-th, and pass B collects the candidates below D* plus the lowest rows at D*.
// A row that is a candidate in several tables is counted in the first one only (its keys of the earlier tables are re-checked:
// 12 B per row).  The <= 32 survivors are sorted across the lanes.  Output: the k-NN key lists of knn.hip.h, so the vote and
// everything after it run unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "geom.h"
#include "knn.hip.h"

namespace slideo {

constexpr int LSH_TIE_CAP = 1024;       // rows at the k-th distance kept in LDS; beyond it the lowest rows are found by repeated minimum passes
constexpr int LSH_MAX_PROBES = 1 + 16 + 120;


// grid ceil(nq_grid / 4), block 256 = 4 waves; q: [nq][8] u32, t: [M][8] u32; out: [nq][KLIST] keys ascending (KNN_EMPTY padding).
// nq_dev != null: the query count lives on the device.
template <int KLIST>
__global__ __launch_bounds__(256) void knn_lsh_kernel(LshDev L, const uint32_t* __restrict__ q, int nq, const uint32_t* __restrict__ t,
                                                      uint32_t* __restrict__ out, const uint32_t* __restrict__ nq_dev) {
    __shared__ uint32_t s_hist[4][260];
    __shared__ uint32_t s_tie[4][LSH_TIE_CAP];
    __shared__ uint32_t s_sel[4][KLIST];
    __shared__ uint32_t s_cnt[4][4];                    // selected, ties, (spare)
    if (nq_dev) nq = (int)*nq_dev;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= nq) return;                               // (wave-uniform; only wave-level synchronisation below)
    uint32_t qd[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qd[i] = __builtin_amdgcn_readfirstlane((int)q[(size_t)qi * 8 + i]);
    const int ntab = L.p.ntab, kb = L.p.kb, mp = L.p.mp;
    uint32_t qk[LSH_MAX_TABLES];
#pragma unroll
    for (int tb = 0; tb < LSH_MAX_TABLES; ++tb) {
        uint32_t k = 0;
        if (tb < ntab)
            for (int b = 0; b < kb; ++b) { const int pos = L.p.bit[tb][b]; k |= ((qd[pos >> 5] >> (pos & 31)) & 1u) << b; }
        qk[tb] = k;
    }
    const int nprobe = mp == 0 ? 1 : (mp == 1 ? 1 + kb : 1 + kb + kb * (kb - 1) / 2);
    auto probe_mask = [&](int pi) -> uint32_t {         // 0, then the single bits, then the pairs (a < b)
        if (pi == 0) return 0u;
        if (pi <= kb) return 1u << (pi - 1);
        int r = pi - 1 - kb, a = 0;
        while (r >= kb - 1 - a) { r -= kb - 1 - a; ++a; }
        return (1u << a) | (1u << (a + 1 + r));
    };
    // every unique candidate once: fn(row, distance)
    auto for_each_candidate = [&](auto&& fn) {
        for (int tb = 0; tb < ntab; ++tb) {
            const int32_t* ofs = L.ofs + (size_t)tb * (L.nbuckets + 1);
            const int32_t* rows = L.rows + (size_t)tb * L.M;
            for (int pi = 0; pi < nprobe; ++pi) {
                const uint32_t b = qk[tb] ^ probe_mask(pi);
                const int lo = ofs[b], hi = ofs[b + 1];
                for (int j = lo + lane; j < hi; j += 64) {
                    const int row = rows[j];
                    bool dup = false;                    // already a candidate through an earlier table?
                    if (tb > 0) {
                        const uint16_t* rk = L.keys + (size_t)row * ntab;
                        for (int e = 0; e < tb; ++e) dup = dup || __popc((uint32_t)rk[e] ^ qk[e]) <= mp;
                    }
                    if (dup) continue;
                    const uint4* tr = reinterpret_cast<const uint4*>(t + (size_t)row * 8);
                    const uint4 a = tr[0], c = tr[1];
                    const int d = __popc(a.x ^ qd[0]) + __popc(a.y ^ qd[1]) + __popc(a.z ^ qd[2]) + __popc(a.w ^ qd[3]) +
                                  __popc(c.x ^ qd[4]) + __popc(c.y ^ qd[5]) + __popc(c.z ^ qd[6]) + __popc(c.w ^ qd[7]);
                    fn(row, d);
                }
            }
        }
    };
    auto wsync = [] { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    for (int i = lane; i < 260; i += 64) s_hist[wave][i] = 0;
    if (lane < 4) s_cnt[wave][lane] = 0;
    wsync();
    // pass A: distance histogram of the candidates
    for_each_candidate([&](int, int d) { atomicAdd(&s_hist[wave][d], 1u); });
    wsync();
    // D* = the smallest distance whose cumulative count reaches KLIST (or 257: fewer candidates than that); below = count(d < D*)
    int dstar = 257; uint32_t below = 0;
    {
        uint32_t run = 0;
        for (int base = 0; base < 257 && dstar == 257; base += 64) {
            const int d = base + lane;
            const uint32_t c = d < 257 ? s_hist[wave][d] : 0u;
            uint32_t inc = c;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) { const uint32_t o = __shfl_up(inc, s); if (lane >= s) inc += o; }
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(run + inc >= (uint32_t)KLIST);
            if (hit) {
                const int l0 = __builtin_ctzll(hit);
                dstar = base + l0;
                below = run + __shfl(inc - c, l0);
            } else run += __shfl(inc, 63);
        }
        if (dstar == 257) below = run;
    }
    const uint32_t need = dstar == 257 ? 0u : (uint32_t)KLIST - below;       // rows to take at distance D*
    // pass B: everything below D*, and the rows at D*
    for_each_candidate([&](int row, int d) {
        if (d < dstar) { const uint32_t s = atomicAdd(&s_cnt[wave][0], 1u); s_sel[wave][s] = ((uint32_t)d << KNN_KEY_SHIFT) | (uint32_t)row; }
        else if (d == dstar) { const uint32_t s = atomicAdd(&s_cnt[wave][1], 1u); if (s < (uint32_t)LSH_TIE_CAP) s_tie[wave][s] = (uint32_t)row; }
    });
    wsync();
    uint32_t nsel = s_cnt[wave][0];
    const uint32_t nties = s_cnt[wave][1];
    if (need > 0) {
        if (nties <= (uint32_t)LSH_TIE_CAP) {
            // the `need` lowest rows of the tie list: rank by counting (list of a few entries almost always)
            for (uint32_t i = lane; i < nties; i += 64) {
                const uint32_t r = s_tie[wave][i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < nties; ++j) rank += s_tie[wave][j] < r ? 1u : 0u;
                if (rank < need) s_sel[wave][nsel + rank] = ((uint32_t)dstar << KNN_KEY_SHIFT) | r;
            }
        } else {
            // more rows at D* than the list holds (a descriptor repeated on > 1000 pages): the lowest rows one minimum pass each
            uint32_t last = 0; bool first = true;
            for (uint32_t got = 0; got < need; ++got) {
                uint32_t mn = 0xFFFFFFFFu;
                for_each_candidate([&](int row, int d) { if (d == dstar && (first || (uint32_t)row > last)) mn = min(mn, (uint32_t)row); });
#pragma unroll
                for (int s = 32; s > 0; s >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, s));
                if (lane == 0) s_sel[wave][nsel + got] = ((uint32_t)dstar << KNN_KEY_SHIFT) | mn;
                last = mn; first = false;
            }
        }
        nsel += need;
    }
    wsync();
    // sort the survivors across the lanes (bitonic, 64 slots) and write the list
    uint32_t key = lane < (int)nsel ? s_sel[wave][lane] : KNN_EMPTY;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)key, j);
            const bool up = (lane & k) == 0, lower = (lane & j) == 0;
            key = (lower == up) ? min(key, o) : max(key, o);
        }
    if (lane < KLIST) out[(size_t)qi * KLIST + lane] = key;
}

// keys of the queries under the tables of L: qkeys [nq][ntab] u16 (what KtHammingLsh::accept compares the rows' keys with).
// grid ceil(nq_grid / 256); nq_dev != null: the query count lives on the device.
__global__ __launch_bounds__(256) void lsh_query_keys_kernel(LshParams P, const uint32_t* __restrict__ q, int nq, uint16_t* __restrict__ qkeys,
                                                             const uint32_t* __restrict__ nq_dev) {
    if (nq_dev) nq = (int)*nq_dev;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    uint32_t d[8];
    {
        const uint4* p = reinterpret_cast<const uint4*>(q + (size_t)i * 8);
        const uint4 a = p[0], b = p[1];
        d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
    }
    for (int tb = 0; tb < P.ntab; ++tb) {
        uint32_t k = 0;
        for (int b = 0; b < P.kb; ++b) {
            const int pos = P.bit[tb][b];
            uint32_t w = d[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) w = (pos >> 5) == e ? d[e] : w;
            k |= ((w >> (pos & 31)) & 1u) << b;
        }
        qkeys[(size_t)i * P.ntab + tb] = (uint16_t)k;
    }
}

}  // namespace slideo
