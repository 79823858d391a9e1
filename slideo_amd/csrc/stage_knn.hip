// stage_knn.hip — index build and the k-NN stage (kernels: knn.hip.h, knn_tile.hip.h, knn_l2.hip.h, knn_lsh.hip.h).
#include "runtime.hpp"
#include "knn.hip.h"
#include "knn_tile.hip.h"
#include "knn_tile1.hip.h"
#include "knn_l2.hip.h"
#include "knn_lsh.hip.h"

using namespace slideo;

namespace slideo {

// ---- exact Hamming kNN: keys into S.d_keys[0 .. nq*KLIST) ------------------------------
static int knn_pad_rows(int nt) { return cdiv(std::max(nt, 1), KT_ST_ROWS) * KT_ST_ROWS; }

// Operand of the {0,1} x {0,1} engine (knn_tile.hip.h): rows in ascending popcount order (stable counting sort on the host:
// nt x 32 bytes of popcounts), expanded to tile-major FP4 on the device, plus per super-tile the rows' norms and original
// indices and per tile half its smallest norm.  `t_host`: the packed rows in host memory.
struct TrainBits { DevBuf *tx, *side, *nminh, *perm; };
// rowid (may be null): the row number a key carries for row i of t_host / t_dev (de-duplicated sets: the lowest original row)
static void prepare_train_bits(const uint8_t* t_host, const uint32_t* t_dev, int nt, TrainBits o, hipStream_t st, const int32_t* rowid = nullptr) {
    const int nt_pad = knn_pad_rows(nt), n_st = nt_pad / KT_ST_ROWS;
    std::vector<uint16_t> norm((size_t)std::max(nt, 1));
    uint32_t hist[258] = {0};
    for (int i = 0; i < nt; ++i) {
        uint64_t w[4];
        std::memcpy(w, t_host + (size_t)i * 32, 32);
        const int n = __builtin_popcountll(w[0]) + __builtin_popcountll(w[1]) + __builtin_popcountll(w[2]) + __builtin_popcountll(w[3]);
        norm[i] = (uint16_t)n; hist[n + 1]++;
    }
    for (int i = 0; i < 257; ++i) hist[i + 1] += hist[i];
    std::vector<int32_t> perm((size_t)nt_pad, -1);
    for (int i = 0; i < nt; ++i) perm[hist[norm[i]]++] = i;             // stable: ties keep row order
    {
        // The 32-row TILES (each of one norm, which is all the fast path needs) are then put in a fixed pseudo-random order.
        // Streaming them in norm order is adversarial for the running thresholds: E[d] = |q| + |t| (1 - |q| / 128), so for every
        // query with more than 128 set bits the nearest rows would come LAST, the k-th distance would keep falling along the
        // stream and almost every tile would send some lane to the slow path (measured: 17.3 ms against 11.9 ms for the
        // +-1 engine on the headline launch).  A shuffled tile order makes the stream i.i.d. again for every query.
        const int ntiles = cdiv(nt, 32);
        std::vector<int32_t> order((size_t)ntiles), shuffled((size_t)nt_pad, -1);
        for (int i = 0; i < ntiles; ++i) order[i] = i;
        uint64_t st_ = 0x9E3779B97F4A7C15ull;
        for (int i = ntiles - 1; i > 0; --i) {
            st_ = st_ * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(order[i], order[(int)((st_ >> 33) % (uint64_t)(i + 1))]);
        }
        for (int p = 0; p < ntiles; ++p)
            for (int r = 0; r < 32; ++r) shuffled[(size_t)p * 32 + r] = perm[(size_t)order[p] * 32 + r];   // (perm is -1 past nt: pad rows)
        perm.swap(shuffled);
    }
    std::vector<uint32_t> side((size_t)n_st * KT_SIDE_U32);
    std::vector<float> nminh((size_t)n_st * 4);
    for (int r = 0; r < nt_pad; ++r) {
        const float nf = perm[r] >= 0 ? (float)norm[perm[r]] : KT_PAD_NORM;     // (pad rows may now sit inside the stream: the partial last tile)
        uint32_t bits; std::memcpy(&bits, &nf, 4);
        side[(size_t)(r / KT_ST_ROWS) * KT_SIDE_U32 + (r % KT_ST_ROWS)] = bits;
        side[(size_t)(r / KT_ST_ROWS) * KT_SIDE_U32 + KT_ST_ROWS + (r % KT_ST_ROWS)] = (uint32_t)(perm[r] >= 0 && rowid ? rowid[perm[r]] : perm[r]);
        if (r % 32 == 0) nminh[r / 32] = 0.5f * nf;                      // ascending order: a tile's first row has its smallest norm
    }
    o.tx->reserve((size_t)nt_pad * 128); o.side->reserve(side.size() * 4 + 16); o.nminh->reserve(nminh.size() * 4 + 16);
    o.perm->reserve(perm.size() * 4 + 16);
    HIP_CHECK(hipMemcpyAsync(o.perm->p, perm.data(), perm.size() * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(o.side->p, side.data(), side.size() * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(o.nminh->p, nminh.data(), nminh.size() * 4, hipMemcpyHostToDevice, st));
    knn_tile_expand_kernel<<<cdiv(nt_pad * 8, 256), 256, 0, st>>>(t_dev, nt, nt_pad, o.perm->as<int32_t>(), o.tx->as<uint4>());
    check_launch("knn_tile_expand_kernel");
    HIP_CHECK(hipStreamSynchronize(st));                                 // the host vectors die here
}

// slideo_config.matcher 1: the tables of FLANN's LshIndex over `nt` host rows (geom.h lsh_params / lsh_key_host), uploaded
static void build_lsh_set(const slideo_config& c, const uint8_t* t_host, int nt, slideo_matcher::LshSet& S, hipStream_t st) {
    const LshParams P = lsh_params(c);
    const int nb = 1 << P.kb;
    std::vector<uint16_t> keys((size_t)std::max(nt, 1) * P.ntab);
    std::vector<int32_t> ofs((size_t)P.ntab * (nb + 1), 0), rows((size_t)P.ntab * std::max(nt, 1));
    for (int tb = 0; tb < P.ntab; ++tb) {
        int32_t* o = ofs.data() + (size_t)tb * (nb + 1);
        for (int i = 0; i < nt; ++i) { const uint32_t k = lsh_key_host(P, tb, t_host + (size_t)i * 32); keys[(size_t)i * P.ntab + tb] = (uint16_t)k; o[k + 1]++; }
        for (int i = 0; i < nb; ++i) o[i + 1] += o[i];
        std::vector<int32_t> cur(o, o + nb);
        for (int i = 0; i < nt; ++i) rows[(size_t)tb * std::max(nt, 1) + cur[keys[(size_t)i * P.ntab + tb]]++] = i;      // rows ascend inside a bucket
    }
    S.ofs.reserve(ofs.size() * 4 + 16); S.rows.reserve(rows.size() * 4 + 16); S.keys.reserve(keys.size() * 2 + 16);
    HIP_CHECK(hipMemcpyAsync(S.ofs.p, ofs.data(), ofs.size() * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(S.rows.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(S.keys.p, keys.data(), keys.size() * 2, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
    S.dev.p = P; S.dev.nbuckets = nb; S.dev.ofs = S.ofs.as<int32_t>(); S.dev.rows = S.rows.as<int32_t>(); S.dev.keys = S.keys.as<uint16_t>();
    S.dev.M = std::max(nt, 1);
    S.ready = true;
}

// One search block per CU instead of two (knn_tile.hip.h: 2 instead of 4 waves per SIMD) while other units are in flight: two
// blocks hold every register of a CU (4 waves x 128 per SIMD), so the ORB and verify kernels of the other units cannot share
// a CU with them and run in the gaps the search launches leave.  With one block the other half of the registers and 72 KB of
// LDS stay free; the search alone is slower (9.4 instead of 7.9 ms for the headline launch), the step of overlapped units
// is 5 % shorter (profiles/r04_experiments.txt).  The block count is capped through the launch's dynamic LDS size: the
// kernel's own 70 KB + this pad exceed half of the CU's 160 KB.  The exact Hamming search only: the LSH-filtered stream is
// VALU-bound on its row filter and dominates its step (one block per CU: 26.2 instead of 21.4 ms per step), and the SIFT
// matcher's step is its extraction stage (L2 search one or two blocks per CU: 73.3 ms either way).
#ifndef KT_SHARE_PAD_V
#define KT_SHARE_PAD_V (16 * 1024)
#endif
constexpr unsigned KT_SHARE_PAD = KT_SHARE_PAD_V;
static_assert(KT_SHARE_PAD == 0 || KT_RING * (KT_ST_U4 * 16 + KT_SIDE_U32 * 4) + KT_SHARE_PAD > 160 * 1024 / 2, "the pad must push a block past half of the CU's LDS");
#ifdef KT_PROBE
void knn_probe_report() {
    unsigned long long v[8] = {0};
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(slideo::kt_probe), sizeof(v)) != hipSuccess || !v[6]) return;
    const double w = (double)v[6];
    fprintf(stderr, "KT_PROBE waves %.0f  cycles per wave: total %.0f  vmcnt %.0f  wait_done %.0f  wait_filled %.0f  slow %.0f  flush %.0f\n",
            w, v[0] / w, v[1] / w, v[2] / w, v[3] / w, v[4] / w, v[5] / w);
    fprintf(stderr, "KT_PROBE s_memtime ticks per 100 MHz wall tick: %.3f (x 100 MHz = the shader clock the waves saw, if s_memtime counts it)\n", (double)v[0] / (double)std::max<unsigned long long>(v[7], 1));
    static unsigned int ww[64][8][4];
    if (hipMemcpyFromSymbol(ww, HIP_SYMBOL(slideo::kt_wave), sizeof(ww)) == hipSuccess)
        for (int b = 0; b < 64; b += 7) {
            fprintf(stderr, "KT_WAVE block %3d:", 100 + b);
            for (int k = 0; k < 8; ++k) fprintf(stderr, "  [%u s%u d%u f%u]", ww[b][k][0], ww[b][k][1], ww[b][k][2], ww[b][k][3]);
            fprintf(stderr, "\n");
        }
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(slideo::kt_probe), z, sizeof(z));
}
#endif
static unsigned share_pad(const slideo_matcher* m, const Slot& S) { return (m->knn_share == 1 || ((m->knn_share < 0 || m->knn_share >= 5) && S.u_shared)) ? KT_SHARE_PAD : 0u; }
// The block shape of the exact Hamming search (engine 3).  SHAPE_T2: knn_tile2_kernel, 8 waves x 2 query tiles, two blocks per CU or
// — with the LDS pad — one.  SHAPE_T2W12 (by default for LARGE decks while units share the chip — knn_w12_ratio —; SLIDEO_KNN_SHARE=3 / 4 force it):
// the same wave shape, 12 waves, one block per CU by its registers.  SHAPE_T1W12 (knn_tile1.hip.h; SLIDEO_KNN_SHARE=5: while other units are in flight, 6: always): 12 waves x 1 query
// tile at 80 registers — three waves per SIMD in the registers two 2-tile waves take; one block per CU by the LDS pad while units
// share the chip, two otherwise.  Only the exact search (matcher 0): the LSH-filtered stream has its own kernel and plan.
enum KnnShape { SHAPE_T2 = 0, SHAPE_T2W12 = 1, SHAPE_T1W12 = 2 };
static KnnShape knn_shape(const slideo_matcher* m, const Slot& S) {
    if (m->cfg.matcher != 0) return SHAPE_T2;
    if ((m->knn_share == 3 && S.u_shared) || m->knn_share == 4) return SHAPE_T2W12;
    if (m->knn_share < 0 && S.u_shared && S.u_w12) return SHAPE_T2W12;       // large decks (runtime.hpp knn_w12_ratio)
    if ((m->knn_share == 5 && S.u_shared) || m->knn_share == 6) return SHAPE_T1W12;
    return SHAPE_T2;
}
static int shape_qpb(KnnShape sh) { return sh == SHAPE_T1W12 ? KT1_QPB : sh == SHAPE_T2W12 ? knn_qpb<2, KT_WAVES12>() : knn_qpb<2>(); }
static int shape_waves(KnnShape sh) { return sh == SHAPE_T1W12 ? KT1_WAVES : sh == SHAPE_T2W12 ? KT_WAVES12 : KT_WAVES; }
static size_t shape_pend_words(KnnShape sh) { return sh == SHAPE_T1W12 ? KT1_PEND_WORDS_PER_WAVE : knn_pend_words_per_wave<2>(); }
struct KnnPlan { int engine, qblocks, nseg, per_seg; };
// Engine 0 ("mfma") = the 2-tile wave shape (knn_tile2_kernel: 4 waves/SIMD, two 512-query blocks per CU) at every size: since the
// {0,1} operand alphabet it runs the headline launch in 10.0 ms alone against 11.3 for the 4-tile shape and the step is 2 %
// shorter.  The 4-tile shape (engine 2) and the VALU popcount kernel (engine 1) stay selectable for A/B: identical results.
static int knn_engine_for(const slideo_matcher* m, int nq) {
    if (m->knn_engine != 0) return m->knn_engine;
    (void)nq;
    return 3;
}
// nq: the query count the plan is made for (the real one, or its estimate when only the device knows it); nq_grid >= nq:
// what the grid and the buffers are sized for (blocks past the device-side count leave at once)
static KnnPlan knn_plan(const slideo_matcher* m, int nq, int nt, int nq_grid = 0, KnnShape sh = SHAPE_T2) {
    KnnPlan p{};
    nq = std::max(nq, 1);            // a unit may hold no keypoint at all (e.g. one flat frame)
    nq_grid = std::max(nq_grid, nq);
    p.engine = knn_engine_for(m, nq);
    if (p.engine == 2 && nt > 0) {
        // one block of 1024 queries per CU: split the train set when fewer query blocks than 3/4 of the CUs exist
        p.qblocks = cdiv(nq, knn_qpb<4>());
        const int n_st = knn_pad_rows(nt) / KT_ST_ROWS;
        int nseg = p.qblocks >= 192 ? 1 : std::min(std::max(256 / std::max(p.qblocks, 1), 1), n_st);
        p.per_seg = cdiv(n_st, std::max(nseg, 1));
        p.nseg = cdiv(n_st, p.per_seg);
        p.qblocks = cdiv(nq_grid, knn_qpb<4>());
    } else if (p.engine == 3 && nt > 0) {
        const int qpb = shape_qpb(sh);
        const bool w12 = sh == SHAPE_T2W12;                              // (one block per CU whatever else runs: 256 slots)
        p.qblocks = cdiv(nq, qpb);
        const int n_st = knn_pad_rows(nt) / KT_ST_ROWS;
        // the chip holds 512 blocks (two per CU).  From 3/4 of that on, one pass over the train set is best (every
        // segment pays its own list warm-up and the merge); fewer query blocks split the train set so that the blocks
        // fill the chip in ONE round (floor, not ceil: 1.4 rounds of smaller blocks lose more to the tail than the
        // empty slots do).  Measured (r01): 236 query blocks x 1.8 M rows (64 4K frames): 1 segment 33.2 ms, 2 segments 24.2 ms,
        // 3 segments 23.5 ms; 239 query blocks x 517 k rows (128 1080p frames): 2 segments 6.24 ms, 3 segments 6.65 ms
        int nseg = p.qblocks >= (w12 ? 192 : 384) ? 1 : std::min(std::max((w12 ? 256 : 512) / std::max(p.qblocks, 1), 1), n_st);
        if (m->knn_nseg_force > 0) nseg = std::min(m->knn_nseg_force, n_st);      // (SLIDEO_KNN_NSEG: measurement)
        p.per_seg = cdiv(n_st, std::max(nseg, 1));
        p.nseg = cdiv(n_st, p.per_seg);
        p.qblocks = cdiv(nq_grid, qpb);
    } else {
        p.engine = 1;
        p.qblocks = cdiv(nq, KNN_BLOCK);
        int nseg = 1;
        if (p.qblocks < 1024) nseg = std::min(cdiv(1024, p.qblocks), std::max(1, nt / 4096));
        p.nseg = std::max(1, std::min(nseg, 256));
        p.per_seg = cdiv(std::max(nt, 1), p.nseg);
        p.qblocks = cdiv(nq_grid, KNN_BLOCK);
    }
    return p;
}

static void knn_reserve(slideo_matcher* m, Slot& S, int nq, int nt, int nq_grid = 0) {
    // (always for the plan that is launched: knn_shape is SHAPE_T2 for the LSH-filtered stream, whose launch plans without a shape)
    const KnnShape sh = knn_shape(m, S);
    const KnnPlan p = knn_plan(m, nq, nt, nq_grid, sh);
    S.d_keys.reserve((size_t)p.nseg * std::max(std::max(nq, nq_grid), 1) * KLIST * 4);
    if (p.engine == 2) S.d_knn_pend.reserve((size_t)p.qblocks * p.nseg * KT_WAVES * knn_pend_words_per_wave<4>() * 4);
    if (p.engine == 3) S.d_knn_pend.reserve((size_t)p.qblocks * p.nseg * shape_waves(sh) * shape_pend_words(sh) * 4);
}

// prune_tol > 0: only neighbours that can pass the vote's `d < best * tol` need to be exact (matrix-core engine; the VALU
// engine always returns full lists)
struct TrainOps {             // device operands of one train set, per engine
    const uint32_t* t;        // packed [nt][8] (VALU engine)
    const uint4* txb;         // {0,1} FP4 expansion in norm order + side arrays (knn_tile.hip.h)
    const uint32_t* side;
    const float4* nminh;
};

// nq_dev != null: the real query count lives on the device (the host did not wait for the ORB counts); then `nq` is the
// estimate the plan is made for and nq_grid the capacity the grid and the buffers cover.  Only the matrix-core engine.
static void run_knn(slideo_matcher* m, Slot& S, const uint32_t* q_dev, int nq, const TrainOps& T, int nt, float prune_tol,
             const uint32_t* nq_dev = nullptr, int nq_grid = 0, hipStream_t st_arg = nullptr) {
    if (nq <= 0 && !nq_dev) return;
    hipStream_t st = st_arg ? st_arg : S.st;
    if ((int64_t)nt >= ((int64_t)1 << KNN_KEY_SHIFT)) fail(SLIDEO_ERR_UNSUPPORTED, "train set of %d rows exceeds %d", nt, 1 << KNN_KEY_SHIFT);
    const KnnShape sh = knn_shape(m, S);
    const KnnPlan p = knn_plan(m, nq, nt, nq_grid, sh);
    knn_reserve(m, S, nq, nt, nq_grid);
    const int nq_all = std::max(std::max(nq, nq_grid), 1);
    unsigned long long* const clk = m->profiling ? m->d_clk.as<unsigned long long>() : nullptr;      // (slideo_matcher_read_shader_clock)
    if ((p.engine == 2 || p.engine == 3) && nt > 0) {
        if (p.engine == 2)
            knn_tile4_kernel<<<dim3(p.qblocks, p.nseg), KT_THREADS, 0, st>>>(q_dev, nq, T.txb, T.side, T.nminh, knn_pad_rows(nt), p.per_seg,
                                                                             S.d_keys.as<uint32_t>(), S.d_knn_pend.as<uint32_t>(), prune_tol, nq_dev);
        else if (sh == SHAPE_T1W12)
            knn_tile1w12_kernel<<<dim3(p.qblocks, p.nseg), KT1_WAVES * 64, 0, st>>>(q_dev, nq, T.txb, T.side, T.nminh, knn_pad_rows(nt), p.per_seg,
                                                                             S.d_keys.as<uint32_t>(), S.d_knn_pend.as<uint32_t>(), prune_tol, nq_dev, clk);
        else if (sh == SHAPE_T2W12)
            knn_tile2w12_kernel<<<dim3(p.qblocks, p.nseg), KT_WAVES12 * 64, 0, st>>>(q_dev, nq, T.txb, T.side, T.nminh, knn_pad_rows(nt), p.per_seg,
                                                                             S.d_keys.as<uint32_t>(), S.d_knn_pend.as<uint32_t>(), prune_tol, nq_dev, clk);
        else
            knn_tile2_kernel<<<dim3(p.qblocks, p.nseg), KT_THREADS, share_pad(m, S), st>>>(q_dev, nq, T.txb, T.side, T.nminh, knn_pad_rows(nt), p.per_seg,
                                                                             S.d_keys.as<uint32_t>(), S.d_knn_pend.as<uint32_t>(), prune_tol, nq_dev, clk);
        check_launch("knn_tile_kernel");
        if (p.nseg > 1) {
            knn_merge_kernel<KLIST><<<cdiv(nq_all, KNN_BLOCK), KNN_BLOCK, 0, st>>>(S.d_keys.as<uint32_t>(), nq, p.nseg, nq_dev);
            check_launch("knn_merge_kernel");
        }
        return;
    }
    if (nq_dev) fail(SLIDEO_ERR_STATE, "internal: the VALU kNN engine needs the query count on the host");
    knn_hamming_kernel<KLIST><<<dim3(p.qblocks, p.nseg), KNN_BLOCK, 0, st>>>(q_dev, nq, T.t, nt, p.per_seg, S.d_keys.as<uint32_t>());
    check_launch("knn_hamming_kernel");
    if (p.nseg > 1) {
        knn_merge_kernel<KLIST><<<p.qblocks, KNN_BLOCK, 0, st>>>(S.d_keys.as<uint32_t>(), nq, p.nseg);
        check_launch("knn_merge_kernel");
    }
}

bool knn_unit_is_valu(const slideo_matcher* m, int nq) { return knn_engine_for(m, nq) == 1; }

// (the matrix-core engine searches the unique rows of the train set, the VALU engine — A/B only — all of them)
static bool knn_unit_dedup(const slideo_matcher* m, int nq) { return m->Mu < m->M && knn_engine_for(m, nq) != 1; }
int knn_unit_rows(const slideo_matcher* m, int nq) { return (int)(knn_unit_dedup(m, nq) ? m->Mu : m->M); }

void knn_reserve_unit(slideo_matcher* m, Slot& S, uint32_t qplan, uint32_t qtot) {
    knn_reserve(m, S, (int)qplan, knn_unit_rows(m, (int)qplan), (int)qtot);
}

// A unit's search: S.d_desc (n frames' descriptors, offsets S.d_qofs) -> S.d_keys.  async: the real query count lives on the
// device (S.d_qofs[n]), qplan is what the launch is planned for and qtot the capacity the grid covers.
void unit_knn(slideo_matcher* m, Slot& S, int n, uint32_t qplan, uint32_t qtot, bool async, bool prof, hipStream_t st) {
    const slideo_config& c = m->cfg;
    const bool dedup = knn_unit_dedup(m, (int)qplan);
    const int nt_knn = knn_unit_rows(m, (int)qplan);
    // a neighbour counts iff d < best * vote_tolerance (verify.hip.h vote_kernel); with tolerance < 1 rows below the
    // current best must still be kept, hence max(tol, 1)
    // (the ratio test needs the exact two nearest rows: exact lists)
    const float prune = (m->knn_exact_lists || m->cfg.ratio_test > 0.f) ? 0.f : std::max(m->cfg.vote_tolerance, 1.0f);
    const TrainOps T{m->d_train.as<uint32_t>(), m->d_trainb.as<uint4>(), m->d_train_side.as<uint32_t>(), m->d_train_nminh.as<float4>()};
    if (c.matcher == 1 && (m->lsh_gather || knn_engine_for(m, (int)qplan) == 1)) {
        // the reference's index, gathered: only the LSH candidates of a query are scored (knn_lsh.hip.h); same key lists out
        knn_lsh_kernel<KLIST><<<cdiv((int)std::max(qtot, 1u), 4), 256, 0, st>>>(m->lsh.dev, S.d_desc.as<uint32_t>(), (int)qtot, m->d_train.as<uint32_t>(),
                                                                               S.d_keys.as<uint32_t>(), async ? S.d_qofs.as<uint32_t>() + n : nullptr);
        check_launch("knn_lsh_kernel");
    } else if (c.matcher == 1) {
        // the same result from the matrix-core stream over ALL rows with the candidate rule applied where a row passes the
        // distance test (KtHammingLsh): a fifth of all rows are candidates of a query on these descriptors (skewed buckets),
        // so gathering them is 60x slower than streaming everything
        const uint32_t* nqd = async ? S.d_qofs.as<uint32_t>() + n : nullptr;
        S.d_qkeys.reserve(std::max<size_t>((size_t)qtot * c.lsh_tables * 2, 64));
        lsh_query_keys_kernel<<<cdiv((int)std::max(qtot, 1u), 256), 256, 0, st>>>(m->lsh.dev.p, S.d_desc.as<uint32_t>(), (int)qtot, S.d_qkeys.as<uint16_t>(), nqd);
        check_launch("lsh_query_keys_kernel");
        const KnnPlan p = knn_plan(m, (int)qplan, nt_knn, (int)qtot);
        const KtLshCtx ctx{m->lsh.dev.keys, S.d_qkeys.as<uint16_t>(), c.lsh_tables, c.lsh_multi_probe};
        knn_tile2_lsh_kernel<<<dim3(p.qblocks, p.nseg), KT_THREADS, 0, st>>>(S.d_desc.as<uint32_t>(), (int)qplan, T.txb, T.side, T.nminh, knn_pad_rows(nt_knn), p.per_seg,
                                                                             S.d_keys.as<uint32_t>(), S.d_knn_pend.as<uint32_t>(), prune, nqd, ctx);
        check_launch("knn_tile2_lsh_kernel");
        if (p.nseg > 1) {
            knn_merge_kernel<KLIST><<<cdiv((int)std::max(qtot, 1u), KNN_BLOCK), KNN_BLOCK, 0, st>>>(S.d_keys.as<uint32_t>(), (int)qplan, p.nseg, nqd);
            check_launch("knn_merge_kernel");
        }
    } else
        run_knn(m, S, S.d_desc.as<uint32_t>(), (int)qplan, T, nt_knn, prune, async ? S.d_qofs.as<uint32_t>() + n : nullptr, (int)qtot, st);
    if (prof) HIP_CHECK(hipEventRecord(S.ev[2], st));      // the kNN interval ends here: the search kernel (+ its segment merge)
    if (dedup) {
        knn_expand_dups_kernel<KLIST><<<cdiv((int)std::max(qtot, 1u), KNN_BLOCK), KNN_BLOCK, 0, st>>>(
            S.d_keys.as<uint32_t>(), (int)qtot, m->d_grp_next.as<int32_t>(), async ? S.d_qofs.as<uint32_t>() + n : nullptr);
        check_launch("knn_expand_dups_kernel");
    }
}

// FlannMatcher::new (mo/flann.rs:65-71) for the Hamming index over the M packed rows of `train` (page order).
void knn_build_index(slideo_matcher* m, const std::vector<uint8_t>& train, int64_t M) {
    m->d_train.reserve(std::max<size_t>(train.size(), 64) + 64);   // + slack: the kNN loop reads whole rows only, no overrun
    HIP_CHECK(hipMemcpy(m->d_train.p, train.data(), train.size(), hipMemcpyHostToDevice));
    // equal rows: sort the row numbers by descriptor (ties by row), chain each group, keep the lowest row of each
    std::vector<int32_t> order((size_t)M), grp_next((size_t)M, -1), urow;
    for (int64_t i = 0; i < M; ++i) order[i] = (int32_t)i;
    m->Mu = M;
    if (m->cfg.matcher == 1) build_lsh_set(m->cfg, train.data(), (int)M, m->lsh, m->stream);
    if (m->knn_dedup && m->cfg.matcher == 0) {
        const uint64_t* t64 = reinterpret_cast<const uint64_t*>(train.data());
        auto less = [&](int32_t a, int32_t b) {
            const uint64_t* x = t64 + (size_t)a * 4; const uint64_t* y = t64 + (size_t)b * 4;
            for (int j = 0; j < 4; ++j) if (x[j] != y[j]) return x[j] < y[j];
            return a < b;
        };
        std::sort(order.begin(), order.end(), less);
        std::vector<uint8_t> head((size_t)M, 1);
        for (int64_t i = 1; i < M; ++i)
            if (std::memcmp(t64 + (size_t)order[i - 1] * 4, t64 + (size_t)order[i] * 4, 32) == 0) { grp_next[order[i - 1]] = order[i]; head[order[i]] = 0; }
        for (int64_t i = 0; i < M; ++i) if (head[i]) urow.push_back((int32_t)i);
        m->Mu = (int64_t)urow.size();
    }
    m->d_grp_next.reserve(grp_next.size() * 4 + 16);
    HIP_CHECK(hipMemcpy(m->d_grp_next.p, grp_next.data(), grp_next.size() * 4, hipMemcpyHostToDevice));
    if (m->Mu < M) {
        std::vector<uint8_t> utrain((size_t)m->Mu * 32);
        for (int64_t i = 0; i < m->Mu; ++i) std::memcpy(utrain.data() + (size_t)i * 32, train.data() + (size_t)urow[i] * 32, 32);
        m->d_utrain.reserve(utrain.size() + 64);
        HIP_CHECK(hipMemcpy(m->d_utrain.p, utrain.data(), utrain.size(), hipMemcpyHostToDevice));
        prepare_train_bits(utrain.data(), m->d_utrain.as<uint32_t>(), (int)m->Mu, TrainBits{&m->d_trainb, &m->d_train_side, &m->d_train_nminh, &m->d_train_perm},
                           m->stream, urow.data());
    } else {
        prepare_train_bits(train.data(), m->d_train.as<uint32_t>(), (int)M, TrainBits{&m->d_trainb, &m->d_train_side, &m->d_train_nminh, &m->d_train_perm}, m->stream);
    }
    HIP_CHECK(hipStreamSynchronize(m->stream));
}

// ---- L2 k-NN (cfg2): train set prepared once, queries from device memory ----
void l2_prepare(slideo_matcher::L2Set& L, const uint8_t* t, int nt, hipStream_t st) {
    const int nt_pad = knn_pad_rows(nt);
    DevBuf d_t, d_norm;
    d_t.reserve(std::max<size_t>((size_t)nt * 128, 64)); d_norm.reserve(std::max<size_t>((size_t)nt * 4, 64));
    L.d_tx.reserve((size_t)nt_pad * 128); L.d_perm.reserve((size_t)nt_pad * 4);
    // norms on the device, the norm order on the host (a stable index sort), then the centred tile-major operand gathered in
    // that order
    std::vector<int32_t> h_norm((size_t)std::max(nt, 1)), h_perm((size_t)nt_pad, -1);
    if (nt) {
        HIP_CHECK(hipMemcpyAsync(d_t.p, t, (size_t)nt * 128, hipMemcpyHostToDevice, st));
        knl_norms_kernel<<<cdiv(nt, 256), 256, 0, st>>>(d_t.as<uint8_t>(), nt, d_norm.as<int32_t>());
        check_launch("knl_norms_kernel");
        HIP_CHECK(hipMemcpyAsync(h_norm.data(), d_norm.p, (size_t)nt * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        for (int i = 0; i < nt; ++i) h_perm[i] = i;
        std::stable_sort(h_perm.begin(), h_perm.begin() + nt, [&](int32_t a, int32_t b) { return h_norm[a] < h_norm[b]; });
        // the 32-row tiles (each of nearly one norm, which is all the fast path needs) in a fixed pseudo-random order: streamed in
        // norm order a query meets its neighbours — rows of about its own norm — only at its own place in the stream and keeps a
        // loose threshold until then (the Hamming engine's finding, prepare_train_bits)
        const int ntiles = cdiv(nt, 32);
        std::vector<int32_t> order((size_t)ntiles), shuffled((size_t)nt_pad, -1);
        for (int i = 0; i < ntiles; ++i) order[i] = i;
        uint64_t st_ = 0x9E3779B97F4A7C15ull;
        for (int i = ntiles - 1; i > 0; --i) {
            st_ = st_ * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(order[i], order[(int)((st_ >> 33) % (uint64_t)(i + 1))]);
        }
        for (int p = 0; p < ntiles; ++p)
            for (int r = 0; r < 32; ++r) shuffled[(size_t)p * 32 + r] = h_perm[(size_t)order[p] * 32 + r];
        h_perm.swap(shuffled);
    }
    // the side array of the tile engine (knn_tile.hip.h): per super-tile 128 negated norms (as i32) and 128 original rows; and
    // per tile the negated norm of its first row (the tile's bound: rows ascend inside a tile)
    const int n_st = nt_pad / KT_ST_ROWS;
    std::vector<uint32_t> side((size_t)n_st * KT_SIDE_U32), tnorm((size_t)n_st * 4);
    for (int r = 0; r < nt_pad; ++r) {
        const int32_t nn = h_perm[r] >= 0 ? -h_norm[h_perm[r]] : -KNL_PAD_NORM;
        side[(size_t)(r / KT_ST_ROWS) * KT_SIDE_U32 + (r % KT_ST_ROWS)] = (uint32_t)nn;
        side[(size_t)(r / KT_ST_ROWS) * KT_SIDE_U32 + KT_ST_ROWS + (r % KT_ST_ROWS)] = (uint32_t)h_perm[r];
        if (r % 32 == 0) tnorm[r / 32] = (uint32_t)nn;
    }
    L.d_side.reserve(side.size() * 4 + 16); L.d_tn.reserve(tnorm.size() * 4 + 16);
    HIP_CHECK(hipMemcpyAsync(L.d_side.p, side.data(), side.size() * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(L.d_tn.p, tnorm.data(), tnorm.size() * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(L.d_perm.p, h_perm.data(), (size_t)nt_pad * 4, hipMemcpyHostToDevice, st));
    knl_expand_train_kernel<<<cdiv(nt_pad * 8, 256), 256, 0, st>>>(d_t.as<uint8_t>(), nt_pad, L.d_perm.as<int32_t>(), L.d_tx.as<uint4>());
    check_launch("knl_expand_train_kernel");
    HIP_CHECK(hipStreamSynchronize(st));            // d_t / d_norm / the host vectors go out of scope
    L.nt = nt; L.nt_pad = nt_pad; L.ready = true;
}

// queries on the device -> idx / dist on the device (m->d_tapidx / d_tapdist); kernel time between two events if asked for
// keys / pend: the list and pending-key buffers of this search — the set's own by default (results then unpacked into
// m->d_tapidx / d_tapdist), a slot's in SIFT matcher mode (the lists are consumed as they are: no unpack)
void l2_query(slideo_matcher* m, slideo_matcher::L2Set& L, const uint8_t* q_dev, int nq, int k, hipStream_t st, Slot& S, bool timed,
              DevBuf* keys, DevBuf* pend, float prune_tol) {
    const int qblocks = cdiv(nq, knn_qpb<2>());
    const bool own = keys == nullptr;
    if (own) { keys = &L.d_keys; pend = &L.d_pend; }
    keys->reserve((size_t)nq * KLIST * 8); pend->reserve((size_t)qblocks * KT_WAVES * knn_pend_words_per_wave<2>() * 8);   // (u64 keys)
    if (own) { m->d_tapidx.reserve((size_t)nq * k * 4); m->d_tapdist.reserve((size_t)nq * k * 4); }
    if (timed) HIP_CHECK(hipEventRecord(S.ev[0], st));
    const int kl = k <= 8 ? 8 : (k <= 16 ? 16 : KLIST);      // list length of the kernel instance (see knn_l2.hip.h)
    if (kl == 8)
        knn_l2_kernel<8><<<qblocks, KT_THREADS, 0, st>>>(q_dev, nq, L.d_tx.as<uint4>(), L.d_side.as<uint32_t>(), L.d_tn.as<uint4>(), L.nt_pad,
                                                          keys->as<unsigned long long>(), pend->as<unsigned long long>(), prune_tol);
    else if (kl == 16)
        knn_l2_kernel<16><<<qblocks, KT_THREADS, 0, st>>>(q_dev, nq, L.d_tx.as<uint4>(), L.d_side.as<uint32_t>(), L.d_tn.as<uint4>(), L.nt_pad,
                                                           keys->as<unsigned long long>(), pend->as<unsigned long long>(), prune_tol);
    else
        knn_l2_kernel<KLIST><<<qblocks, KT_THREADS, 0, st>>>(q_dev, nq, L.d_tx.as<uint4>(), L.d_side.as<uint32_t>(), L.d_tn.as<uint4>(), L.nt_pad,
                                                              keys->as<unsigned long long>(), pend->as<unsigned long long>(), prune_tol);
    check_launch("knn_l2_kernel");
    if (own) {
        knl_unpack_kernel<<<cdiv(nq * k, 256), 256, 0, st>>>(keys->as<unsigned long long>(), nq, kl, k, m->d_tapidx.as<int32_t>(), m->d_tapdist.as<uint32_t>());
        check_launch("knl_unpack_kernel");
    }
    if (timed) HIP_CHECK(hipEventRecord(S.ev[1], st));
}

// SIFT matcher mode: the outcome of the vote rule on the L2 lists (u64 keys, kl per query) as neighbour lists in the HAMMING key
// format, so that the vote kernel and everything after it run unchanged (knn_l2.hip.h l2_ratio_keys_kernel / l2_tol_keys_kernel)
void l2_lists_to_keys(slideo_matcher* m, Slot& S, const DevBuf& lists, int kq, uint32_t qtot, bool lowe, hipStream_t st) {
    const int kl = kq <= 8 ? 8 : (kq <= 16 ? 16 : KLIST);                 // (the list length of the instance l2_query picked)
    if (lowe)
        l2_ratio_keys_kernel<<<cdiv((int)qtot, 256), 256, 0, st>>>(lists.as<unsigned long long>(), kl, (int)qtot, m->sift_ratio, S.d_keys.as<uint32_t>(), KLIST);
    else
        l2_tol_keys_kernel<<<cdiv((int)qtot, 256), 256, 0, st>>>(lists.as<unsigned long long>(), kl, kq, (int)qtot, m->cfg.vote_tolerance, S.d_keys.as<uint32_t>(), KLIST);
    check_launch("l2 keys kernel");
}

}  // namespace slideo

extern "C" {

int32_t slideo_knn_hamming(slideo_matcher* m, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, int32_t k,
                           int32_t* idx_out, uint16_t* dist_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (nq < 0 || nt < 0 || k < 1 || k > KLIST) fail(SLIDEO_ERR_INVALID_ARG, "bad nq/nt/k (k must be 1..%d)", KLIST);
    if ((nq && !q) || (nt && !t) || (nq && (!idx_out || !dist_out))) fail(SLIDEO_ERR_INVALID_ARG, "null argument");
    if (nq == 0) return SLIDEO_OK;
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    hipStream_t st = S.st;
    m->d_tapq.reserve((size_t)nq * 32); m->d_tapt.reserve(std::max<size_t>((size_t)nt * 32, 64));
    HIP_CHECK(hipMemcpyAsync(m->d_tapq.p, q, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    if (nt) HIP_CHECK(hipMemcpyAsync(m->d_tapt.p, t, (size_t)nt * 32, hipMemcpyHostToDevice, st));
    DevBuf tapb, tap_side, tap_nminh, tap_perm;
    if (knn_engine_for(m, nq) != 1 && nt > 0) prepare_train_bits(t, m->d_tapt.as<uint32_t>(), nt, TrainBits{&tapb, &tap_side, &tap_nminh, &tap_perm}, st);
    run_knn(m, S, m->d_tapq.as<uint32_t>(), nq, TrainOps{m->d_tapt.as<uint32_t>(), tapb.as<uint4>(), tap_side.as<uint32_t>(), tap_nminh.as<float4>()}, nt, 0.f);
    m->d_tapidx.reserve((size_t)nq * k * 4); m->d_tapdist.reserve((size_t)nq * k * 2);
    knn_unpack_kernel<<<cdiv(nq * k, 256), 256, 0, st>>>(S.d_keys.as<uint32_t>(), nq, KLIST, k, m->d_tapidx.as<int32_t>(), m->d_tapdist.as<uint16_t>());
    check_launch("knn_unpack_kernel");
    HIP_CHECK(hipMemcpyAsync(idx_out, m->d_tapidx.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(dist_out, m->d_tapdist.p, (size_t)nq * k * 2, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    API_CATCH(m)
}

int32_t slideo_knn_lsh(slideo_matcher* m, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, int32_t k, int32_t* idx_out, uint16_t* dist_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (nq < 0 || nt < 0 || k < 1 || k > KLIST) fail(SLIDEO_ERR_INVALID_ARG, "bad nq/nt/k (k must be 1..%d)", KLIST);
    if ((nq && !q) || (nt && !t) || (nq && (!idx_out || !dist_out))) fail(SLIDEO_ERR_INVALID_ARG, "null argument");
    if ((int64_t)nt >= ((int64_t)1 << KNN_KEY_SHIFT)) fail(SLIDEO_ERR_UNSUPPORTED, "train set of %d rows exceeds %d", nt, 1 << KNN_KEY_SHIFT);
    if (m->cfg.lsh_tables < 1 || m->cfg.lsh_tables > 8 || m->cfg.lsh_key_bits < 1 || m->cfg.lsh_key_bits > 16 || m->cfg.lsh_multi_probe < 0 || m->cfg.lsh_multi_probe > 2)
        fail(SLIDEO_ERR_UNSUPPORTED, "lsh_tables must be 1..8, lsh_key_bits 1..16, lsh_multi_probe 0..2");
    if (nq == 0) return SLIDEO_OK;
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    hipStream_t st = S.st;
    slideo_matcher::LshSet set;
    build_lsh_set(m->cfg, t, nt, set, st);
    m->d_tapq.reserve((size_t)nq * 32); m->d_tapt.reserve(std::max<size_t>((size_t)nt * 32, 64) + 64);
    HIP_CHECK(hipMemcpyAsync(m->d_tapq.p, q, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    if (nt) HIP_CHECK(hipMemcpyAsync(m->d_tapt.p, t, (size_t)nt * 32, hipMemcpyHostToDevice, st));
    S.d_keys.reserve((size_t)nq * KLIST * 4);
    knn_lsh_kernel<KLIST><<<cdiv(nq, 4), 256, 0, st>>>(set.dev, m->d_tapq.as<uint32_t>(), nq, m->d_tapt.as<uint32_t>(), S.d_keys.as<uint32_t>(), nullptr);
    check_launch("knn_lsh_kernel");
    m->d_tapidx.reserve((size_t)nq * k * 4); m->d_tapdist.reserve((size_t)nq * k * 2);
    knn_unpack_kernel<<<cdiv(nq * k, 256), 256, 0, st>>>(S.d_keys.as<uint32_t>(), nq, KLIST, k, m->d_tapidx.as<int32_t>(), m->d_tapdist.as<uint16_t>());
    check_launch("knn_unpack_kernel");
    HIP_CHECK(hipMemcpyAsync(idx_out, m->d_tapidx.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(dist_out, m->d_tapdist.p, (size_t)nq * k * 2, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    API_CATCH(m)
}

int32_t slideo_l2_set_train(slideo_matcher* m, const uint8_t* t, int32_t nt) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (nt < 0 || (nt && !t)) fail(SLIDEO_ERR_INVALID_ARG, "null train set");
    if (m->sift_on) fail(SLIDEO_ERR_STATE, "the L2 train set is the page DB's in SIFT mode");
    if ((int64_t)nt >= ((int64_t)1 << KNN_KEY_SHIFT)) fail(SLIDEO_ERR_UNSUPPORTED, "train set of %d rows exceeds %d", nt, 1 << KNN_KEY_SHIFT);
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    l2_prepare(m->l2, t, nt, m->slots[0].st);
    API_CATCH(m)
}

int32_t slideo_l2_knn_dev(slideo_matcher* m, const void* q_dev, int32_t nq, int32_t k, void* idx_dev, void* dist_dev, float* kernel_ms) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (!m->l2.ready) fail(SLIDEO_ERR_STATE, "slideo_l2_set_train must be called first");
    if (nq < 0 || k < 1 || k > KLIST) fail(SLIDEO_ERR_INVALID_ARG, "bad nq/k (k must be 1..%d)", KLIST);
    if (nq && (!q_dev || !idx_dev || !dist_dev)) fail(SLIDEO_ERR_INVALID_ARG, "null argument");
    if (kernel_ms) *kernel_ms = 0.f;
    if (nq == 0) return SLIDEO_OK;
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    l2_query(m, m->l2, static_cast<const uint8_t*>(q_dev), nq, k, S.st, S, kernel_ms != nullptr);
    HIP_CHECK(hipMemcpyAsync(idx_dev, m->d_tapidx.p, (size_t)nq * k * 4, hipMemcpyDeviceToDevice, S.st));
    HIP_CHECK(hipMemcpyAsync(dist_dev, m->d_tapdist.p, (size_t)nq * k * 4, hipMemcpyDeviceToDevice, S.st));
    HIP_CHECK(hipStreamSynchronize(S.st));
    if (kernel_ms) HIP_CHECK(hipEventElapsedTime(kernel_ms, S.ev[0], S.ev[1]));
    API_CATCH(m)
}

int32_t slideo_knn_l2_u8(slideo_matcher* m, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, int32_t k,
                         int32_t* idx_out, uint32_t* dist_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (nq < 0 || nt < 0 || k < 1 || k > KLIST) fail(SLIDEO_ERR_INVALID_ARG, "bad nq/nt/k (k must be 1..%d)", KLIST);
    if ((nq && !q) || (nt && !t) || (nq && (!idx_out || !dist_out))) fail(SLIDEO_ERR_INVALID_ARG, "null argument");
    if ((int64_t)nt >= ((int64_t)1 << KNN_KEY_SHIFT)) fail(SLIDEO_ERR_UNSUPPORTED, "train set of %d rows exceeds %d", nt, 1 << KNN_KEY_SHIFT);
    if (nq == 0) return SLIDEO_OK;
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    hipStream_t st = S.st;
    slideo_matcher::L2Set tap;                    // a set of its own: the one installed by slideo_l2_set_train stays as it is
    l2_prepare(tap, t, nt, st);
    DevBuf d_q;
    d_q.reserve((size_t)nq * 128);
    HIP_CHECK(hipMemcpyAsync(d_q.p, q, (size_t)nq * 128, hipMemcpyHostToDevice, st));
    l2_query(m, tap, d_q.as<uint8_t>(), nq, k, st, S, false);
    HIP_CHECK(hipMemcpyAsync(idx_out, m->d_tapidx.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(dist_out, m->d_tapdist.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    API_CATCH(m)
}

}  // extern "C"
