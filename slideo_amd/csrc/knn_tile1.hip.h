// knn_tile1.hip.h — the exact Hamming k-NN engine of knn_tile.hip.h as a block of THREE waves per SIMD that fits the registers
// of two: ONE 32-query tile and ONE accumulator per wave (88 registers), twelve waves, 384 queries per block.
// MEASUREMENT MODE (SLIDEO_KNN_SHARE=5 / 6): built, held bit-identical by the tests, measured — and slower than the shipped shape
// (profiles/r06_experiments.txt 1: 96 against 116 queries per ms and CU); kept because the numbers in DESIGN.md section 8 come from it.
//
// Why.  While units share the chip the search runs one block per CU and leaves the other half of every SIMD's registers to the
// ORB / verify kernels of the other units (stage_knn.hip).  With the 2-tile wave shape that block is two waves per SIMD at 128
// registers each, and two waves keep a SIMD's matrix pipe 56 - 62 % busy where three or four reach 90 % (profiles/r05_experiments.txt
// 8): the pipe idles whenever both waves are between MFMAs.  Round 5's twelve-wave block of the SAME wave shape filled the pipe but
// took 384 of the 512 registers per SIMD lane and lost what it had gained to the co-runners' occupancy (experiments 3, 10).
// This shape pays for the third wave with registers the 2-tile wave spends on instruction-level parallelism, which three
// waves per SIMD do not need:
//   * one query tile per wave: 16 registers of B operand instead of 32;
//   * ONE accumulator, no skew: the max tree of a tile runs when its four MFMAs are through — the other two waves of the SIMD
//     own the pipe meanwhile (a wave needs ~90 cycles between its last MFMA and its next first one, the others have 256 to issue);
//   * one fragment set, each k-step's registers re-loaded for the next tile as soon as its MFMA has issued.
// 3 x 88 = 264 registers per SIMD lane against the 256 of two 2-tile waves: the co-runners keep their occupancy.
// The price is LDS traffic — a fragment read feeds one MFMA instead of two: 50 % of the LDS's ds_read_b128 rate at a full pipe —
// and 4/3 of the L2 -> LDS staging per query (384 instead of 512 queries per streamed matrix).
// One block per CU by its LDS (the ring below: 88 KB).  Same contract, ring, push / flush protocol, thresholds, fused vote
// filter and result as knn_tile_body<2, KtHamming>; tests/test_gpu_parity.py holds all engines to the same keys.
#pragma once
#include "knn_tile.hip.h"

namespace slideo {

constexpr int KT1_WAVES = 12;
// The ring of this shape.  A super-tile lasts a wave 16 MFMAs (~1500 cycles with three waves on the pipe) — less than an LDS-DMA
// fetch takes to land once the other units' kernels keep evicting the matrix from the L2 (it comes from the Infinity Cache then:
// > 1 us under load).  knn_tile_body publishes a share one step after it was issued, behind `s_waitcnt vmcnt(0)`: one super-tile
// time as the landing deadline, which stalled this shape at EVERY super-tile (first build: block life 10 ms against 2.3 ms of
// MFMAs).  Here a share is published KT1_LAG steps after its issue, behind `s_waitcnt vmcnt(loads issued since)` (vector-memory
// loads return in order), and is issued KT1_AHEAD = KT1_LAG + 1 steps before its first reader; KT1_RING = KT1_AHEAD + 2 slots.
#ifndef KT1_LAG_V
#define KT1_LAG_V 2
#endif
constexpr int KT1_LAG = KT1_LAG_V, KT1_AHEAD = KT1_LAG + 1, KT1_RING = KT1_AHEAD + 2;
constexpr int KT1_LOADS_PER_STAGE = KT_ST_U4 / KT_WAVES / 64;    // operand pieces of a staging wave (waves 0..3 also issue one for the side array: the wait below is then stricter than needed)
constexpr int KT1_QPB = KT1_WAVES * 32;                       // queries per block
constexpr size_t KT1_PEND_WORDS_PER_WAVE = (size_t)KT_PEND_CAP * 64;

template <int W>
__device__ __forceinline__ void knn_tile1_body(const uint8_t* __restrict__ q, int nq, const uint4* __restrict__ tx,
                                               const uint32_t* __restrict__ side, const uint4* __restrict__ nminh, int nt_pad,
                                               int st_per_seg, uint32_t* __restrict__ out, uint32_t* __restrict__ pend_ws,
                                               float prune_tol, const uint32_t* __restrict__ nq_dev, unsigned long long* __restrict__ clk) {
    typedef KtHamming M;
    typedef M::Acc Acc;
    constexpr int KL = M::KL;
    static_assert(W % 4 == 0 && W >= KT_WAVES, "waves w, w + 4, ... share a SIMD; waves 0 .. KT_WAVES - 1 stage the ring");
    if (nq_dev) nq = (int)*nq_dev;                                     // (capacity-sized grid: knn_tile_body)
    if ((int)blockIdx.x * (W * 32) >= nq) return;
    kt_clock(clk, false);
#ifdef KT_PROBE
    unsigned long long kt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    KT_T0(kt_all);
    const unsigned long long kt_wall = wall_clock64();
#endif
    constexpr int RING = KT1_RING, AHEAD = KT1_AHEAD, LAG = KT1_LAG;
    __shared__ uint4 lds[RING][KT_ST_U4];
    __shared__ __attribute__((aligned(16))) uint32_t lds_side[RING][KT_SIDE_U32];
    __shared__ uint32_t s_filled[RING], s_done[RING];
    __shared__ float s_nq[W][32];                                      // |q| of every query of the block (slow path and flush only)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, ql = lane & 31;
    const int qbase = blockIdx.x * (W * 32) + wave * 32;
    const int seg = blockIdx.y;
    const int n_st = nt_pad / KT_ST_ROWS;
    const int st0 = seg * st_per_seg, st1 = min(n_st, st0 + st_per_seg);
    const int nst = st1 - st0;
    // Addresses are a wave-uniform base (scalar registers) + a 32-bit per-lane byte offset — a per-lane 64-bit pointer is two vector
    // registers, and this wave shape has 80.
    // this wave's pending keys: [KT_PEND_CAP][64 lanes]; lanes l and l + 32 push for query l (rows 0..15 / 16..31 of a tile's pattern)
    const char* const Pb = reinterpret_cast<const char*>(pend_ws + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * W + wave) * KT1_PEND_WORDS_PER_WAVE);
    // a wave (or lanes) past the last query works on query nq - 1 again and writes nothing
    const int qb = min(qbase, nq - 1);                                 // (wave-uniform)
    const uint32_t qoff = (uint32_t)min(ql, max(nq - 1 - qbase, 0));   // this lane's query = qb + qoff
    const char* const qrow_b = reinterpret_cast<const char*>(q) + (size_t)qb * M::QBYTES;
    char* const list_b = reinterpret_cast<char*>(out + ((size_t)seg * nq + qb) * KL);

    // `opaque(lane)`: a copy of the lane number the optimiser cannot trace — what is computed from it stays where it is written
    // instead of being hoisted out of the main loop into registers that live through it (a per-lane 64-bit global address is two)
    auto opaque = [](uint32_t v) -> uint32_t { return kt_opaque(v); };
    M::Bop bq[4];
    auto load_queries = [&]() {
        const uint32_t lo = opaque((uint32_t)lane) & 31u;
        const uint8_t* qp = reinterpret_cast<const uint8_t*>(qrow_b + (uint32_t)min((int)lo, max(nq - 1 - qbase, 0)) * (uint32_t)M::QBYTES);
        int pc = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) M::load_b(qp, s, half, bq[s], pc);
        const int tot = pc + __shfl_xor(pc, 32);
        if (half == 0) s_nq[wave][ql] = (float)tot;
    };
    load_queries();
    auto nq_of = [&]() -> float { return s_nq[wave][ql]; };
    // the list of query ql of this wave; lane ql owns it (lane ql + 32 computes the same and does not write)
    auto my_list = [&]() -> uint4* { return reinterpret_cast<uint4*>(list_b + qoff * (uint32_t)(KL * 4)); };       // (prologue only)
    auto owner_valid = [&]() -> bool { return half == 0 && qbase + ql < nq; };
    constexpr int LIST_U4 = KL / 4;
    if (owner_valid()) {
        uint4* l = my_list();
#pragma unroll
        for (int i = 0; i < LIST_U4; ++i) l[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    float h = M::open_thr(nq_of());                                   // (|q| - B - 1) / 2, B = the largest distance still accepted
    uint32_t cntp = 0;                                                 // keys pending in this lane's column of P

    auto flush = [&]() {
        const uint32_t c_lo = __shfl(cntp, ql), c_hi = __shfl(cntp, ql + 32);
        const uint32_t lo = opaque((uint32_t)lane) & 31u;              // (= ql, recomputed here: see opaque)
        const char* PP = Pb + 4u * lo;
        uint4* const ml = reinterpret_cast<uint4*>(list_b + (uint32_t)min((int)lo, max(nq - 1 - qbase, 0)) * (uint32_t)(KL * 4));
        uint32_t lst[KL];
#pragma unroll
        for (int i = 0; i < LIST_U4; ++i) {
            const uint4 v = ml[i];
            lst[4 * i] = v.x; lst[4 * i + 1] = v.y; lst[4 * i + 2] = v.z; lst[4 * i + 3] = v.w;
        }
        for (uint32_t base = 0; __builtin_amdgcn_ballot_w64(base < c_lo) != 0ull; base += KT_FLUSH_BATCH) {
            uint32_t e[KT_FLUSH_BATCH];
#pragma unroll
            for (int i = 0; i < KT_FLUSH_BATCH; ++i) e[i] = base + i < c_lo ? *reinterpret_cast<const uint32_t*>(PP + (base + i) * 256u) : ~0u;
#pragma unroll
            for (int i = 0; i < KT_FLUSH_BATCH; ++i) M::insert(lst, e[i]);
        }
        for (uint32_t base = 0; __builtin_amdgcn_ballot_w64(base < c_hi) != 0ull; base += KT_FLUSH_BATCH) {
            uint32_t e[KT_FLUSH_BATCH];
#pragma unroll
            for (int i = 0; i < KT_FLUSH_BATCH; ++i) e[i] = base + i < c_hi ? *reinterpret_cast<const uint32_t*>(PP + (base + i) * 256u + 128u) : ~0u;
#pragma unroll
            for (int i = 0; i < KT_FLUSH_BATCH; ++i) M::insert(lst, e[i]);
        }
        if (owner_valid()) {
#pragma unroll
            for (int i = 0; i < LIST_U4; ++i) ml[i] = make_uint4(lst[4 * i], lst[4 * i + 1], lst[4 * i + 2], lst[4 * i + 3]);
        }
        // the query's new bound (both half-waves hold the same list): the k-th distance, inclusive (knn_tile.hip.h header), and
        // with prune_tol the vote's acceptance bound d <= ceil(best * tol) - 1
        float bnd = lst[31] == KNN_EMPTY ? 512.f : (float)(lst[31] >> KNN_KEY_SHIFT);
        if (prune_tol > 0.f) bnd = fminf(bnd, ceilf((float)(lst[0] >> KNN_KEY_SHIFT) * prune_tol) - 1.f);
        h = (nq_of() - bnd - 1.f) * 0.5f;
        cntp = 0;
    };

    // ---- LDS ring: knn_tile_body's, W readers, waves 0 .. KT_WAVES - 1 stage -----------------------------------------------
    auto stage = [&](int jj, int sl) {
        constexpr int PER_WAVE = KT_ST_U4 / KT_WAVES;
        const uint32_t lo = opaque((uint32_t)lane);
        const char* sb = reinterpret_cast<const char*>(tx + (size_t)(st0 + jj) * KT_ST_U4 + wave * PER_WAVE);     // (wave-uniform)
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
    auto signal = [&](uint32_t* f) { kt_ring_signal(f, lane); };       // (the counters in inline assembly: knn_tile.hip.h kt_ring_peek)
    auto wait_ge = [&](uint32_t* f, uint32_t target) { kt_ring_wait_ge(f, target); };
    if (tid < RING) { s_filled[tid] = 0; s_done[tid] = 0; }
    __syncthreads();
    const bool stager = wave < KT_WAVES;
    if (stager) {
        for (int j = 0; j < AHEAD && j < nst; ++j) stage(j, j);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int j = 0; j < AHEAD && j < nst; ++j) signal(&s_filled[j]);
    }
    // this wave is about to read super-tile j (called one tile before): issue the share of super-tile j + AHEAD, publish the one
    // issued LAG calls ago (super-tile j + 1), wait for super-tile j
    auto acquire = [&](int j, uint32_t e_done, uint32_t e_filled) {
        const int jn = j + AHEAD, jp = j + AHEAD - LAG;
        kt_ring_landed();                                              // (the early peeks, issued a tile ago)
        if (stager && jn < nst) {                                      // the slot was last read for super-tile jn - RING (by all W waves)
            const uint32_t tgt = (uint32_t)W * (uint32_t)(jn / RING);
            KT_T0(t_b);
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)e_done) < tgt) wait_ge(&s_done[jn % RING], tgt);
            KT_T1(2, t_b);
#ifndef KT1_NO_DMA      /* experiment: the stream re-reads the ring's first slots (wrong results): the kernel without its fetches */
            stage(jn, jn % RING);
#endif
        }
        if (stager && j >= LAG && jp < nst) {                          // (super-tiles < AHEAD were published by the prologue)
            // LAG stages were issued behind super-tile jp's while the stream lasts (fewer at its end: then wait for everything)
            KT_T0(t_a);
            if (jn < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LAG * KT1_LOADS_PER_STAGE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            KT_T1(1, t_a);
            signal(&s_filled[jp % RING]);
        }
        const uint32_t tgf = (uint32_t)KT_WAVES * (uint32_t)(j / RING + 1);
        KT_T0(t_c);
#ifndef KT1_NO_DMA
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)e_filled) < tgf) wait_ge(&s_filled[j % RING], tgf);
#endif
        asm volatile("" ::: "memory");
        KT_T1(3, t_c);
    };
    auto peek = [&](const uint32_t* f) -> uint32_t { return kt_ring_peek(f); };

    auto tree = [&](const Acc& acc) -> int {
        int tk[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) tk[k] = max(max(M::raw(acc[3 * k]), M::raw(acc[3 * k + 1])), M::raw(acc[3 * k + 2]));
        return max(max(max(M::raw(acc[15]), tk[0]), tk[1]), max(max(tk[2], tk[3]), tk[4]));
    };
    // slow path: knn_tile_body's (Hamming), one query tile
    auto candidates = [&](const Acc& acc, int thri, const uint32_t* sd, int tt) {
        float dbest = 1024.f;
        const float nq_i = nq_of();
        uint32_t c = cntp;
        const uint32_t lane4 = opaque((uint32_t)lane) * 4u;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int kk = k < 5 ? k : 0;
            const bool gate = (k < 5 ? max(max(M::raw(acc[3 * kk]), M::raw(acc[3 * kk + 1])), M::raw(acc[3 * kk + 2])) : M::raw(acc[15])) > thri;
            if (__builtin_amdgcn_ballot_w64(gate) == 0ull) continue;
            uint32_t nrm3[3], row3[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int r = min(3 * k + u, 15);
                nrm3[u] = sd[tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
                row3[u] = sd[KT_ST_ROWS + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
            }
#pragma unroll
            for (int r = 3 * k; r < 3 * k + 3 && r < 16; ++r) {
                const float nrm = __uint_as_float(nrm3[r - 3 * k]);
                const float v = acc[r];
                const bool hit = __builtin_fmaf(nrm, -0.5f, v) > h;              // exact: halves of small integers
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(hit) != 0ull, 0)) {
                    if (hit) {
                        const float d = nq_i + nrm - 2.f * v;
                        *reinterpret_cast<uint32_t*>(const_cast<char*>(Pb) + (c * 256u + lane4)) = ((uint32_t)(int)d << KNN_KEY_SHIFT) | row3[r - 3 * k];
                        ++c;
                        dbest = fminf(dbest, d);
                    }
                }
            }
        }
        cntp = c;
        if (prune_tol > 0.f) {                                           // fused vote filter, applied at once (knn_tile_body)
            float bn = ceilf(dbest * prune_tol) - 1.f;
            bn = kt_min_halves(bn);
            h = fmaxf(h, (nq_i - bn - 1.f) * 0.5f);
        }
    };

    if (nst > 0) {
        const Acc zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        Acc a;
        // One tile: its four MFMAs, each k-step's fragment registers re-loaded for the NEXT tile (Lx_) behind the MFMA that read
        // them; then the tile's max tree and test.
#ifdef KT1_FENCED      /* experiment: every fragment re-load pinned right behind the MFMA that read its registers */
#define KT1_FENCE __builtin_amdgcn_sched_barrier(0);
#else
#define KT1_FENCE
#endif
#define KT1_TILE(tt)                                                                                                   \
        {                                                                                                             \
            const uint4* Lx_ = (tt) == KT_TPS - 1 ? Ln : Lc + ((tt) + 1) * 256;                                       \
            KT1_FENCE a = M::mfma(zero, f0, bq[0]); KT1_FENCE f0 = Lx_[0];                                            \
            KT1_FENCE a = M::mfma(a, f1, bq[1]); KT1_FENCE f1 = Lx_[64];                                              \
            KT1_FENCE a = M::mfma(a, f2, bq[2]); KT1_FENCE f2 = Lx_[128];                                             \
            KT1_FENCE a = M::mfma(a, f3, bq[3]); KT1_FENCE f3 = Lx_[192]; KT1_FENCE                                   \
            const uint32_t nmh_ = (tt) == 0 ? nm4.x : (tt) == 1 ? nm4.y : (tt) == 2 ? nm4.z : nm4.w;                  \
            const int mx_ = tree(a), ti_ = M::tile_thr(h, nmh_);                                                      \
            if (__builtin_amdgcn_ballot_w64(mx_ > ti_) != 0ull) { KT_T0(t_s); candidates(a, ti_, sdc, tt); KT_T1(4, t_s); }      \
        }
        acquire(0, 0u, 0u);
        const uint4* Lc = lds[0] + lane;
        uint4 f0 = Lc[0], f1 = Lc[64], f2 = Lc[128], f3 = Lc[192];
        uint4 nm4n = nminh[st0];                                       // (fetched one super-tile ahead: knn_tile_body)
#pragma unroll 1
        for (int j = 0; j < nst; ++j) {
            const int slot = j % RING;
            // the waves of a SIMD (w, w + 4, w + 8, ...) lead in turns, one super-tile each (knn_tile_body)
            if ((j + (wave >> 2)) % (W / 4) == 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
            Lc = lds[slot] + lane;
            const uint32_t* sdc = lds_side[slot];
            const uint4 nm4 = nm4n;
            nm4n = nminh[st0 + min(j + 1, nst - 1)];
            const uint4* Ln = j + 1 < nst ? lds[(j + 1) % RING] + lane : Lc + (KT_TPS - 1) * 256;
            KT1_TILE(0)
            KT1_TILE(1)
            uint32_t e_done = 0, e_filled = 0;
            if (j + 1 < nst) {
                e_filled = peek(&s_filled[(j + 1) % RING]);
                if (j + 1 + AHEAD < nst) e_done = peek(&s_done[(j + 1 + AHEAD) % RING]);
            }
            KT1_TILE(2)
            if (j + 1 < nst) acquire(j + 1, e_done, e_filled);         // tile 3 loads the next super-tile's first fragments
            KT1_TILE(3)
            signal(&s_done[slot]);
            if (__builtin_amdgcn_ballot_w64(cntp >= (uint32_t)KT_FLUSH_AT) != 0ull) {
                KT_T0(t_f);
                flush();
                KT_T1(5, t_f);
                load_queries();                                        // (rebuilt, not kept alive across the flush's 32 list registers)
                f0 = Ln[0]; f1 = Ln[64]; f2 = Ln[128]; f3 = Ln[192];
            }
        }
#undef KT1_TILE
#undef KT1_FENCE
    }
    flush();
    kt_clock(clk, true);
#ifdef KT_PROBE
    KT_T1(0, kt_all);
    kt_acc[6] = 1; kt_acc[7] = wall_clock64() - kt_wall;
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&kt_probe[i], kt_acc[i]);
#endif
}

// 88 registers (amdgpu_num_vgpr counts in units of two on gfx90a+): three waves per SIMD in 264 registers beside the other units'
// kernels; ONE block per CU by its LDS (5 ring slots: 88 KB), whatever else runs
__global__ __attribute__((amdgpu_num_vgpr(44))) __launch_bounds__(KT1_WAVES * 64, 3)
void knn_tile1w12_kernel(const uint32_t* __restrict__ q, int nq, const uint4* __restrict__ tx, const uint32_t* __restrict__ side,
                         const float4* __restrict__ nminh, int nt_pad, int st_per_seg, uint32_t* __restrict__ out,
                         uint32_t* __restrict__ pend_ws, float prune_tol, const uint32_t* __restrict__ nq_dev, unsigned long long* __restrict__ clk) {
    knn_tile1_body<KT1_WAVES>(reinterpret_cast<const uint8_t*>(q), nq, tx, side, reinterpret_cast<const uint4*>(nminh), nt_pad, st_per_seg, out, pend_ws,
                              prune_tol, nq_dev, clk);
}

}  // namespace slideo
