// stage_sift.hip — the SIFT stage and its entry points (kernels: sift.hip.h).
#include "runtime.hpp"
#include "sift.hip.h"

using namespace slideo;

namespace slideo {

static SiftGeom sift_geom(int w, int h) {
    SiftGeom g{};
    g.w = w; g.h = h;
    int bw = 2 * w, bh = 2 * h;
    g.n_oct = std::max(0, (int)std::lrint(std::log((double)std::min(bw, bh)) / std::log(2.) - 2) + 1);
    if (g.n_oct > SIFT_MAX_OCT) g.n_oct = SIFT_MAX_OCT;
    int64_t go = 0, dofs = 0;
    for (int o = 0; o < g.n_oct; ++o) {
        g.ow[o] = bw; g.oh[o] = bh; g.g_ofs[o] = go; g.d_ofs[o] = dofs;
        go += (int64_t)bw * bh * (SIFT_NL + 3); dofs += (int64_t)bw * bh * (SIFT_NL + 2);
        bw /= 2; bh /= 2;
        if (bw < 1 || bh < 1) { g.n_oct = o + 1; break; }
    }
    g.g_frame = go; g.d_frame = dofs;
    return g;
}

static SiftTaps sift_taps(double sigma) {          // getGaussianKernel(cvRound(8 sigma + 1) | 1, sigma) in f32 (oracle sift_gauss_kernel)
    SiftTaps t{};
    const int n = (int)std::lrint(sigma * 4 * 2 + 1) | 1;
    if (n > SIFT_MAX_TAPS - 1) fail(SLIDEO_ERR_UNSUPPORTED, "SIFT blur of sigma %.3f needs %d taps (> %d)", sigma, n, SIFT_MAX_TAPS - 1);
    t.n = n;
    double kd[SIFT_MAX_TAPS], sum = 0;
    const double s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; kd[i] = std::exp(s2 * x * x); sum += kd[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) t.k[i] = (float)(kd[i] * sum);
    return t;
}

void sift_check_cfg(const slideo_sift_config* sc, int w, int h) {
    if (!sc) fail(SLIDEO_ERR_INVALID_ARG, "null SIFT config");
    if (sc->n_octave_layers != SIFT_NL) fail(SLIDEO_ERR_UNSUPPORTED, "SIFT n_octave_layers must be %d", SIFT_NL);
    if (!(sc->sigma > 0.5) || !(sc->contrast_threshold >= 0) || !(sc->edge_threshold > 0) || sc->nfeatures < 0) fail(SLIDEO_ERR_INVALID_ARG, "bad SIFT config");
    if (w > 4095 || h > 4095) fail(SLIDEO_ERR_UNSUPPORTED, "SIFT image %dx%d: sides must be <= 4095", w, h);
}

template <int N>
static bool sift_blur_fast(bool fma, const float* src, int64_t sf, float* dst, int64_t df, float* dog, int64_t dgf, int w, int h, int n, const SiftTaps& tp, hipStream_t st,
                    float* half, int64_t hf, int hw, int hh) {      // returns: the half-size copy was written too
    // streams: one wave per (64-column strip, chunk of rows); chunks sized so that a launch has ~16 k waves (several rounds of the chip: a short tail)
    const int strips = cdiv(w, 64);
    constexpr int target_waves = 16384;
    const int chunks = std::min(std::max(cdiv(target_waves, strips * std::max(n, 1)), 1), cdiv(h, 64));
    const int chunk_h = (cdiv(h, chunks) + 7) & ~7;
    const dim3 grid(cdiv(strips * cdiv(h, chunk_h), 4), n);
    if constexpr (N == 17) {            // (the tap count of layer nOctaveLayers at the default sigma: the instance that can write the half-size copy)
        if (half) {
            if (fma) sift_blur_stream_kernel<N, true, true><<<grid, 256, 0, st>>>(src, sf, dst, df, dog, dgf, w, h, tp, chunk_h, half, hf, hw, hh);
            else sift_blur_stream_kernel<N, false, true><<<grid, 256, 0, st>>>(src, sf, dst, df, dog, dgf, w, h, tp, chunk_h, half, hf, hw, hh);
            return true;
        }
    }
    if (fma) sift_blur_stream_kernel<N, true><<<grid, 256, 0, st>>>(src, sf, dst, df, dog, dgf, w, h, tp, chunk_h, nullptr, 0, 0, 0);
    else sift_blur_stream_kernel<N, false><<<grid, 256, 0, st>>>(src, sf, dst, df, dog, dgf, w, h, tp, chunk_h, nullptr, 0, 0, 0);
    return false;
}

// half != null: also leave dst's every-second-pixel copy (hw x hh) there if the kernel taken can (returns whether it did)
static bool sift_blur_launch(slideo_matcher* m, const float* src, int64_t sf, float* dst, int64_t df, float* dog, int64_t dgf, int w, int h, int n,
                      const SiftTaps& tp, hipStream_t st, float* half = nullptr, int64_t hf = 0, int hw = 0, int hh = 0) {
    bool half_done = false;
    const bool fma = m->cfg.ocv.blur != 1;
    bool fast = w >= 32 && h >= 32;
    if (fast) {
        switch (tp.n) {
#define SLIDEO_SIFT_CASE(N) case N: half_done = sift_blur_fast<N>(fma, src, sf, dst, df, dog, dgf, w, h, n, tp, st, half, hf, hw, hh); break;
            SLIDEO_SIFT_CASE(7) SLIDEO_SIFT_CASE(9) SLIDEO_SIFT_CASE(11) SLIDEO_SIFT_CASE(13) SLIDEO_SIFT_CASE(15) SLIDEO_SIFT_CASE(17)
            SLIDEO_SIFT_CASE(19) SLIDEO_SIFT_CASE(21) SLIDEO_SIFT_CASE(23) SLIDEO_SIFT_CASE(25) SLIDEO_SIFT_CASE(27)
#undef SLIDEO_SIFT_CASE
            default: fast = false;
        }
    }
    if (!fast) {
        const dim3 grid(cdiv(w, SIFT_BT_W) * cdiv(h, SIFT_BT_H), n);
        if (fma) sift_blur_kernel<true><<<grid, 256, 0, st>>>(src, sf, dst, df, dog, dgf, w, h, tp);
        else sift_blur_kernel<false><<<grid, 256, 0, st>>>(src, sf, dst, df, dog, dgf, w, h, tp);
    }
    check_launch("sift_blur_kernel");
    return half_done;
}

// Gaussian + DoG pyramids of nb frames (device) into m->sift.gauss / dog
static void sift_pyramids(slideo_matcher* m, const uint8_t* frames_dev, int nb, int w, int h, int stride, int64_t fs, const slideo_sift_config& sc,
                   const SiftGeom& g, hipStream_t st) {
    auto& W = m->sift;
    const int64_t base_frame = (int64_t)g.ow[0] * g.oh[0];
    W.base.reserve((size_t)base_frame * nb * 4);
    W.gauss.reserve((size_t)g.g_frame * nb * 4 + 64);
    {
        // gray u8 first (the ORB path's kernel), then the doubled f32 base image from it (the DoG
        // pyramid itself is not stored: sift.hip.h SiftDog)
        const int gp = ((w + 15) & ~15) + 16;
        const int64_t gframe = (int64_t)gp * h;
        W.gray.reserve((size_t)gframe * nb + 64);
        orb_launch_gray(m, frames_dev, fs, stride, W.gray.as<uint8_t>(), gframe, w, h, gp, nb, st);
        sift_base_kernel<<<dim3(cdiv(w, 512), cdiv(h, SIFT_BASE_ROWS), nb), 256, 0, st>>>(W.gray.as<uint8_t>(), gframe, gp, w, h, W.base.as<float>(), base_frame);
        check_launch("sift_base_kernel");
    }
    const float sigma = (float)sc.sigma;
    const float sig_diff = std::sqrt(std::max(sigma * sigma - 0.5f * 0.5f * 4, 0.01f));
    float* G = W.gauss.as<float>();
    sift_blur_launch(m, W.base.as<float>(), base_frame, G + g.g_ofs[0], g.g_frame, nullptr, 0, g.ow[0], g.oh[0], nb, sift_taps(sig_diff), st);
    double sig[SIFT_NL + 3];
    sig[0] = sc.sigma;
    const double k = std::pow(2., 1. / SIFT_NL);
    for (int i = 1; i < SIFT_NL + 3; ++i) { const double sp = std::pow(k, (double)(i - 1)) * sc.sigma, stt = sp * k; sig[i] = std::sqrt(stt * stt - sp * sp); }
    bool half_done = false;                       // the octave's first layer was written by the previous octave's layer-3 blur
    for (int o = 0; o < g.n_oct; ++o) {
        const int ow = g.ow[o], oh = g.oh[o];
        const int64_t lsz = (int64_t)ow * oh;
        if (o > 0 && !half_done) {
            sift_half_kernel<<<dim3(cdiv(ow, 256), oh, nb), 256, 0, st>>>(G + g.g_ofs[o - 1] + (int64_t)g.ow[o - 1] * g.oh[o - 1] * SIFT_NL, g.g_frame, g.ow[o - 1],
                                                                          G + g.g_ofs[o], g.g_frame, ow, oh);
            check_launch("sift_half_kernel");
        }
        half_done = false;
        for (int i = 1; i < SIFT_NL + 3; ++i) {
            const bool want_half = i == SIFT_NL && o + 1 < g.n_oct;
            const bool did = sift_blur_launch(m, G + g.g_ofs[o] + lsz * (i - 1), g.g_frame, G + g.g_ofs[o] + lsz * i, g.g_frame, nullptr, 0,      // (the DoG layers are not stored: SiftDog)
                                              ow, oh, nb, sift_taps(sig[i]), st,
                                              want_half ? G + g.g_ofs[o + 1] : nullptr, g.g_frame, want_half ? g.ow[o + 1] : 0, want_half ? g.oh[o + 1] : 0);
            if (want_half) half_done = did;
        }
    }
}

// SIFT of n device frames: results appended at kp_dev / desc_dev from row `row0`; per-frame counts into counts_host[0..n).
// Returns the rows written.
static int64_t sift_batch(slideo_matcher* m, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t fs, const slideo_sift_config& sc,
                   int64_t row0, int64_t capacity_total, slideo_keypoint* kp_dev, uint8_t* desc_dev, uint32_t* counts_host, hipStream_t st) {
    auto& W = m->sift;
    const SiftGeom g = sift_geom(w, h);
    SiftParams sp{};
    sp.nfeatures = sc.nfeatures; sp.cand_cap = 1 << 16; sp.raw_cap = 1 << 15;
    if (const long c0 = env_long("SLIDEO_SIFT_LIST_CAP", 0)) {          // (read per call: the tests start small to force the growth path)
        sp.cand_cap = (int)std::min(std::max(c0, 64l), 1l << 16); sp.raw_cap = std::max(sp.cand_cap / 2, 32);
    }
    sp.contrast_threshold = (float)sc.contrast_threshold; sp.edge_threshold = (float)sc.edge_threshold; sp.sigma = (float)sc.sigma;
    sp.threshold = (int)std::floor(0.5 * sc.contrast_threshold / SIFT_NL * 255);
    sp.atan_fma = m->cfg.ocv.atan; sp.blur_fma = m->cfg.ocv.blur != 1;
    // frames per pass under the budget for the pyramids (265 MB + 33 MB per 1080p frame; 96 GB on an MI355X, 24 GB on a small device;
    // SLIDEO_SIFT_WS_MB: tests)
    const size_t per = ((size_t)g.g_frame + (size_t)g.ow[0] * g.oh[0]) * 4;
    size_t budget = (size_t)std::max(env_long("SLIDEO_SIFT_WS_MB", m->sift_ws_mb), 1l) << 20;             // (read per call: a test squeezes it)
    {
        // ... and under what the device can give NOW: half of the free memory plus what the pyramids already hold (a co-tenant —
        // a second matcher on the device, a group with a repeated ordinal, the caller's own tensors — shrinks the pass instead of
        // failing its hipMalloc; the result does not depend on how the batch is cut)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, std::max<size_t>((free_b + W.gauss.cap + W.base.cap) / 2, (size_t)64 << 20));
        else (void)hipGetLastError();
    }
    const int nb_max = (int)std::max<size_t>(1, budget / std::max<size_t>(per, 1));
    int64_t rows = 0;
    for (int f0 = 0; f0 < n; f0 += nb_max) {
        const int nb = std::min(nb_max, n - f0);
        sift_pyramids(m, frames_dev + (int64_t)f0 * fs, nb, w, h, stride, fs, sc, g, st);
        // The extrema and the refined keypoints of a frame go to lists of fixed capacity; a frame beyond one (a very busy 4K page,
        // noise) raises a device flag, and the pass over these pyramids runs again with that list doubled (the pyramids stay).
        uint32_t info[6];
        std::vector<uint32_t> hc((size_t)nb * 3);
        for (;;) {
        W.cand.reserve((size_t)nb * sp.cand_cap * 4);
        W.counts.reserve((size_t)nb * 3 * 4 + 16);                 // cand_count | raw_count | kept_count
        W.raw.reserve((size_t)nb * sp.raw_cap * sizeof(SiftRaw));
        W.items.reserve((size_t)nb * sp.raw_cap * 8);
        W.kept.reserve((size_t)nb * sp.raw_cap * 4);
        W.qofs.reserve((size_t)(nb + 1) * 4); W.info.reserve(64);
        uint32_t* cand_count = W.counts.as<uint32_t>();
        uint32_t* raw_count = cand_count + nb;
        uint32_t* kept_count = raw_count + nb;
        uint32_t* flags = W.info.as<uint32_t>() + 4;
        HIP_CHECK(hipMemsetAsync(W.counts.p, 0, (size_t)nb * 3 * 4, st));
        HIP_CHECK(hipMemsetAsync(W.info.p, 0, 32, st));
        for (int o = 0; o < g.n_oct; ++o) {
            if (g.ow[o] <= 2 * SIFT_BORDER || g.oh[o] <= 2 * SIFT_BORDER) continue;
            sift_extrema_kernel<<<dim3(cdiv(g.ow[o] - 2 * SIFT_BORDER, 4 * SIFT_EX_COLS), cdiv(g.oh[o] - 2 * SIFT_BORDER, SIFT_EX_RCH), nb), 256, 0, st>>>(
                g, sp, o, W.gauss.as<float>(), W.cand.as<uint32_t>(), cand_count, flags);
            check_launch("sift_extrema_kernel");
        }
        // (the candidate count is on the device: the grid covers the capacity, surplus waves leave at once)
        HIP_CHECK(hipMemcpyAsync(hc.data(), W.counts.p, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        uint32_t maxc = 0;
        for (int i = 0; i < nb; ++i) maxc = std::max(maxc, std::min(hc[i], (uint32_t)sp.cand_cap));
        if (maxc > 0) {
            sift_refine_kernel<<<dim3(cdiv((int)maxc, 256), nb), 256, 0, st>>>(g, sp, W.gauss.as<float>(), W.cand.as<uint32_t>(), cand_count,
                                                                            W.raw.as<SiftRaw>(), raw_count, flags);
            check_launch("sift_refine_kernel");
        }
        sift_select_kernel<<<nb, 1024, 0, st>>>(sp, W.raw.as<SiftRaw>(), raw_count, W.items.as<uint64_t>(), W.kept.as<uint32_t>(), kept_count);
        check_launch("sift_select_kernel");
        orb_launch_scan(kept_count, nb, W.qofs.as<uint32_t>(), W.info.as<uint32_t>(), st);
        HIP_CHECK(hipMemcpyAsync(info, W.info.p, 24, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(hc.data(), kept_count, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (info[4] & 16u) {
            if (sp.cand_cap >= (1 << 24)) fail(SLIDEO_ERR_CAPACITY, "SIFT: more than %d scale-space extrema in a frame", sp.cand_cap);
            sp.cand_cap *= 2;
            continue;
        }
        if (info[4] & 32u) {
            // (sift_select_kernel packs a keypoint's slot in 16 bits)
            if (sp.raw_cap >= (1 << 16)) fail(SLIDEO_ERR_CAPACITY, "SIFT: more than %d keypoints in a frame before retainBest", sp.raw_cap);
            sp.raw_cap *= 2;
            continue;
        }
        break;
        }
        const uint32_t total = info[0];
        for (int i = 0; i < nb; ++i) counts_host[f0 + i] = hc[i];
        if (row0 + rows + (int64_t)total > capacity_total) { rows += total; continue; }      // (keeps counting: the caller learns the size it needs)
        if (total > 0) {
            sift_describe_kernel<<<cdiv((int)total, 4), 256, 0, st>>>(g, sp, nb, W.gauss.as<float>(), W.raw.as<SiftRaw>(), W.kept.as<uint32_t>(), W.qofs.as<uint32_t>(),
                                                                      kp_dev + row0 + rows, desc_dev + (size_t)(row0 + rows) * 128);
            check_launch("sift_describe_kernel");
        }
        rows += total;
    }
    return rows;
}

// rows of keypoint / descriptor capacity a SIFT-mode unit of n frames reserves (ties at retainBest's threshold are kept)
static int64_t sift_unit_capacity(const slideo_matcher* m, int n) {
    const int per = m->sift_cfg.nfeatures > 0 ? std::min(m->sift_cfg.nfeatures + 2048, 1 << 15) : (1 << 15);
    return (int64_t)n * per;
}

// slideo_matcher_use_sift: a unit = SIFT on the frames -> squared-L2 k-NN (k = 2) against the deck's SIFT rows -> ratio test as
// Hamming-format neighbour lists (l2_ratio_keys_kernel) -> the common verify stage.  The SIFT workspace belongs to the matcher,
// so the extraction stages of consecutive units take turns (event chain); a unit's search (matrix cores, the slot's own list
// buffers) and verify stages overlap the next unit's extraction.  The keypoint counts come back to the host inside sift_batch: the submit blocks for the extraction.
void unit_submit_sift(slideo_matcher* m, Slot& S, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride) {
    const slideo_config& c = m->cfg;
    hipStream_t st = S.st;
    const bool prof = m->profiling;
    S.timed = prof; S.u_frames = frames_dev; S.u_w = w; S.u_h = h; S.u_stride = stride; S.u_fs = frame_stride; S.u_async = false;
    if (m->sift_ev_set) HIP_CHECK(hipStreamWaitEvent(st, m->sift_ev, 0));
    if (prof) HIP_CHECK(hipEventRecord(S.ev[0], st));
    std::vector<uint32_t> counts((size_t)std::max(n, 1), 0);
    int64_t cap = sift_unit_capacity(m, n), rows = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        S.d_kp.reserve(std::max<size_t>((size_t)cap * sizeof(slideo_keypoint), 64));
        S.d_desc.reserve(std::max<size_t>((size_t)cap * 128, 128));
        rows = sift_batch(m, frames_dev, n, w, h, stride, frame_stride, m->sift_cfg, 0, cap, S.d_kp.as<slideo_keypoint>(), S.d_desc.as<uint8_t>(),
                          counts.data(), st);
        if (rows <= cap) break;
        cap = rows;                                                           // (nfeatures 0 on a very busy frame: once more with room)
    }
    const uint32_t qtot = (uint32_t)rows;
    S.orb.qofs.assign((size_t)n + 1, 0);
    uint32_t mx = 0;
    for (int i = 0; i < n; ++i) { S.orb.qofs[i + 1] = S.orb.qofs[i] + counts[i]; mx = std::max(mx, counts[i]); }
    S.orb.qtot = qtot; S.orb.max_count = mx; S.orb.nframes = n;
    S.u_nt = (int)m->M;
    S.d_qofs.reserve((size_t)(n + 1) * 4 + 16); S.d_info.reserve(16); S.d_flags.reserve(16);
    S.h_info.reserve((size_t)(n + 1) * 4 + 16);
    uint32_t* hq = S.h_info.as<uint32_t>();
    std::memcpy(hq, S.orb.qofs.data(), (size_t)(n + 1) * 4);
    hq[n + 1] = qtot; hq[n + 2] = mx;
    HIP_CHECK(hipMemcpyAsync(S.d_qofs.p, hq, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(S.d_info.p, hq + n + 1, 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemsetAsync(S.d_flags.p, 0, 16, st));
    const bool lowe = m->sift_ratio > 0.f;                                   // ratio test (k = 2) or the path's tolerance vote (k = knn_k)
    const int kq = lowe ? 2 : c.knn_k;
    S.d_keys.reserve(std::max<size_t>((size_t)qtot * KLIST * 4, 64));
    S.d_votes.reserve(std::max<size_t>((size_t)qtot * kq * sizeof(uint2), 16));
    S.d_gpts.reserve(std::max<size_t>((size_t)qtot * kq * sizeof(float4), 16));
    S.d_gmask.reserve(std::max<size_t>((size_t)qtot * kq, 16));
    S.d_fcs.reserve((size_t)n * sizeof(FrameCands));
    S.d_verdicts.reserve((size_t)n * sizeof(slideo_verdict));
    S.d_pairs.reserve((size_t)n * MAXR * sizeof(PairDesc) + 64);
    S.h_out.reserve((size_t)n * (sizeof(slideo_verdict) + sizeof(FrameCands)) + 64);
    HIP_CHECK(hipMemsetAsync(S.d_fcs.p, 0, (size_t)n * sizeof(FrameCands), st));
    HIP_CHECK(hipEventRecord(m->sift_ev, st));                               // the matcher's SIFT workspace is free again: the next unit's
    m->sift_ev_set = true;                                                    // extraction runs beside this unit's search (matrix cores) and verify
    if (prof) HIP_CHECK(hipEventRecord(S.ev[1], st));
    if (qtot > 0) {
        // the search writes this SLOT's list / pending buffers (u64 keys; d_blur is unused in this mode)
        // (tolerance vote: the search keeps its lists exact only for the rows that can pass it — the fused filter of the Hamming engine)
        l2_query(m, m->l2, S.d_desc.as<uint8_t>(), (int)qtot, kq, st, S, false, &S.d_blur, &S.d_knn_pend, !lowe ? c.vote_tolerance : 0.f);
        l2_lists_to_keys(m, S, S.d_blur, kq, qtot, lowe, st);
    }
    if (prof) HIP_CHECK(hipEventRecord(S.ev[2], st));
    VerifyParams vp = make_vp(c);
    vp.rng_len = m->rng_len;
    // (the lists carry the outcome of the vote rule: l2_ratio_keys_kernel / l2_tol_keys_kernel)
    if (lowe) { vp.k = 2; vp.ratio = 1.f; }
    else { vp.k = kq; vp.ratio = 0.f; vp.tol = 1.5f; }
    unit_verify(m, S, vp, frames_dev, n, w, h, stride, frame_stride, qtot);
}

// page ingest in SIFT mode: the staged pages (S.d_stage) -> S.d_kp / S.d_desc (128 B rows) / S.orb.qofs, as run_orb leaves them
void add_pages_sift(slideo_matcher* m, Slot& S, int cnt, int w, int h, int stride, int64_t fb) {
    sift_check_cfg(&m->sift_cfg, w, h);
    std::vector<uint32_t> counts((size_t)cnt, 0);
    int64_t cap = sift_unit_capacity(m, cnt), rows = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        S.d_kp.reserve(std::max<size_t>((size_t)cap * sizeof(slideo_keypoint), 64));
        S.d_desc.reserve(std::max<size_t>((size_t)cap * 128, 128));
        rows = sift_batch(m, S.d_stage.as<uint8_t>(), cnt, w, h, stride, fb, m->sift_cfg, 0, cap, S.d_kp.as<slideo_keypoint>(), S.d_desc.as<uint8_t>(),
                          counts.data(), S.st);
        if (rows <= cap) break;
        cap = rows;
    }
    HIP_CHECK(hipStreamSynchronize(S.st));
    S.orb.qofs.assign((size_t)cnt + 1, 0);
    for (int i = 0; i < cnt; ++i) S.orb.qofs[i + 1] = S.orb.qofs[i] + counts[i];
    S.orb.qtot = (uint32_t)rows; S.orb.nframes = cnt;
}

}  // namespace slideo

extern "C" {

void slideo_sift_config_default(slideo_sift_config* c) {
    if (!c) return;
    c->nfeatures = 0; c->n_octave_layers = 3; c->contrast_threshold = 0.04; c->edge_threshold = 10; c->sigma = 1.6;
}

int32_t slideo_matcher_use_sift(slideo_matcher* m, const slideo_sift_config* cfg, float ratio) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (!m->pages.empty() || m->finalized) fail(SLIDEO_ERR_STATE, "slideo_matcher_use_sift must precede the first page");
    sift_check_cfg(cfg, 64, 64);
    if (!(ratio >= 0.f) || !(ratio <= 1.f)) fail(SLIDEO_ERR_INVALID_ARG, "ratio must be in [0, 1] (0 = the path's tolerance vote)");
    if (m->cfg.matcher != 0) fail(SLIDEO_ERR_UNSUPPORTED, "the LSH index is a Hamming index: not with SIFT features");
    m->sift_on = true; m->sift_cfg = *cfg; m->sift_ratio = ratio;
    API_CATCH(m)
}

int32_t slideo_sift_frames_dev(slideo_matcher* m, const slideo_sift_config* cfg, int32_t n_frames, const uint8_t* frames_dev, int32_t width,
                               int32_t height, int32_t stride_bytes, int64_t frame_stride_bytes, int64_t capacity_total, void* kp_dev,
                               void* desc_dev, uint32_t* qofs_out, float* kernel_ms) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (n_frames < 0 || !qofs_out || (n_frames > 0 && (!frames_dev || !kp_dev || !desc_dev))) fail(SLIDEO_ERR_INVALID_ARG, "null argument");
    validate_image(width, height, stride_bytes);
    sift_check_cfg(cfg, width, height);
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    if (kernel_ms) { *kernel_ms = 0.f; HIP_CHECK(hipEventRecord(S.ev[0], S.st)); }
    std::vector<uint32_t> counts((size_t)std::max(n_frames, 1), 0);
    const int64_t rows = sift_batch(m, frames_dev, n_frames, width, height, stride_bytes, frame_stride_bytes, *cfg, 0, capacity_total,
                                    static_cast<slideo_keypoint*>(kp_dev), static_cast<uint8_t*>(desc_dev), counts.data(), S.st);
    if (kernel_ms) HIP_CHECK(hipEventRecord(S.ev[1], S.st));
    HIP_CHECK(hipStreamSynchronize(S.st));
    if (kernel_ms) HIP_CHECK(hipEventElapsedTime(kernel_ms, S.ev[0], S.ev[1]));
    qofs_out[0] = 0;
    for (int i = 0; i < n_frames; ++i) qofs_out[i + 1] = qofs_out[i] + counts[i];
    if (rows > capacity_total) fail(SLIDEO_ERR_CAPACITY, "SIFT found %lld keypoints, capacity %lld", (long long)rows, (long long)capacity_total);
    API_CATCH(m)
}

int32_t slideo_sift_bgr8(slideo_matcher* m, const slideo_sift_config* cfg, const uint8_t* bgr, int32_t width, int32_t height, int32_t stride_bytes,
                         slideo_keypoint* kp, uint8_t* desc128, int32_t capacity, int32_t* n_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (!bgr || !n_out) fail(SLIDEO_ERR_INVALID_ARG, "null image/n_out");
    validate_image(width, height, stride_bytes);
    sift_check_cfg(cfg, width, height);
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    const size_t fb = (size_t)height * stride_bytes;
    stage_for_upload(m, fb);
    HIP_CHECK(hipMemcpyAsync(S.d_stage.p, bgr, fb, hipMemcpyHostToDevice, S.st));
    const int64_t cap = std::max<int64_t>(capacity, 0);
    m->sift.kp.reserve(std::max<size_t>((size_t)cap * sizeof(slideo_keypoint), 64));
    m->sift.desc.reserve(std::max<size_t>((size_t)cap * 128, 128));
    uint32_t cnt = 0;
    const int64_t rows = sift_batch(m, S.d_stage.as<uint8_t>(), 1, width, height, stride_bytes, (int64_t)fb, *cfg, 0, cap,
                                    m->sift.kp.as<slideo_keypoint>(), m->sift.desc.as<uint8_t>(), &cnt, S.st);
    *n_out = (int32_t)rows;
    if (rows > cap) fail(SLIDEO_ERR_CAPACITY, "image has %lld SIFT keypoints, capacity %d", (long long)rows, capacity);
    if (rows > 0) {
        if (kp) HIP_CHECK(hipMemcpyAsync(kp, m->sift.kp.p, (size_t)rows * sizeof(slideo_keypoint), hipMemcpyDeviceToHost, S.st));
        if (desc128) HIP_CHECK(hipMemcpyAsync(desc128, m->sift.desc.p, (size_t)rows * 128, hipMemcpyDeviceToHost, S.st));
    }
    HIP_CHECK(hipStreamSynchronize(S.st));
    API_CATCH(m)
}

int32_t slideo_sift_layer_bgr8(slideo_matcher* m, const slideo_sift_config* cfg, const uint8_t* bgr, int32_t width, int32_t height, int32_t stride_bytes,
                               int32_t octave, int32_t layer, int32_t dog, float* out, int64_t out_capacity, int32_t* lw, int32_t* lh) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (!bgr || !out || !lw || !lh) fail(SLIDEO_ERR_INVALID_ARG, "null argument");
    validate_image(width, height, stride_bytes);
    sift_check_cfg(cfg, width, height);
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    const SiftGeom g = sift_geom(width, height);
    if (octave < 0 || octave >= g.n_oct || layer < 0 || layer >= (dog ? SIFT_NL + 2 : SIFT_NL + 3)) fail(SLIDEO_ERR_INVALID_ARG, "no such pyramid layer");
    const size_t fb = (size_t)height * stride_bytes;
    stage_for_upload(m, fb);
    HIP_CHECK(hipMemcpyAsync(S.d_stage.p, bgr, fb, hipMemcpyHostToDevice, S.st));
    sift_pyramids(m, S.d_stage.as<uint8_t>(), 1, width, height, stride_bytes, (int64_t)fb, *cfg, g, S.st);
    const int64_t lsz = (int64_t)g.ow[octave] * g.oh[octave];
    *lw = g.ow[octave]; *lh = g.oh[octave];
    if (lsz > out_capacity) fail(SLIDEO_ERR_CAPACITY, "layer has %lld values", (long long)lsz);
    const float* src = m->sift.gauss.as<float>() + g.g_ofs[octave] + lsz * layer;
    HIP_CHECK(hipMemcpyAsync(out, src, (size_t)lsz * 4, hipMemcpyDeviceToHost, S.st));
    std::vector<float> upper;
    if (dog) {              // DoG layer L = Gaussian layer L + 1 - layer L (not stored on the device: sift.hip.h SiftDog)
        upper.resize((size_t)lsz);
        HIP_CHECK(hipMemcpyAsync(upper.data(), src + lsz, (size_t)lsz * 4, hipMemcpyDeviceToHost, S.st));
    }
    HIP_CHECK(hipStreamSynchronize(S.st));
    if (dog) for (int64_t i = 0; i < lsz; ++i) out[i] = upper[(size_t)i] - out[i];
    API_CATCH(m)
}

}  // extern "C"
