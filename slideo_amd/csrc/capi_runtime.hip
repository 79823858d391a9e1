// capi_runtime.hip — handles, page database, workspace slots, unit submit / collect and the match entry points of include/slideo_amd.h (no kernel of its own).
#include "runtime.hpp"
#include <chrono>

using namespace slideo;

namespace slideo {

namespace {
std::string g_create_error;
std::mutex g_err_mutex;
}

void check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) fail(SLIDEO_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
}

GeomEntry& geom_for(slideo_matcher* m, int w, int h) {
    for (auto& g : m->geoms) if (g->w == w && g->h == h) return *g;
    if (w < 1 || h < 1 || w > MAX_DIM || h > MAX_DIM)
        fail(SLIDEO_ERR_UNSUPPORTED, "image size %dx%d outside 1..%d", w, h, MAX_DIM);
    auto e = std::make_unique<GeomEntry>();
    e->w = w; e->h = h;
    std::vector<uint32_t> tab;
    build_pyr_geom(w, h, m->cfg, e->g, tab);
    orb_geom_init(m, *e, tab);
    m->geoms.push_back(std::move(e));
    return *m->geoms.back();
}

int area_class_for(slideo_matcher* m, int w, int h) {
    for (size_t i = 0; i < m->area_geoms.size(); ++i)
        if (m->area_geoms[i].sw == w && m->area_geoms[i].sh == h) return (int)i;
    AreaGeom a;
    if (!build_area_geom(w, h, m->cfg.small_area, a, m->area_taps, m->area_idx, m->cfg.ocv.area, &m->area_recs))
        fail(SLIDEO_ERR_UNSUPPORTED, "image %dx%d has area below small_area=%d: to_small_image would upscale (INTER_AREA falls back to bilinear in OpenCV), not implemented",
             w, h, m->cfg.small_area);
    m->area_geoms.push_back(a);
    m->area_dirty = true;
    return (int)m->area_geoms.size() - 1;
}

void upload_area(slideo_matcher* m) {
    if (!m->area_dirty) return;
    m->d_area_geoms.reserve(m->area_geoms.size() * sizeof(AreaGeom));
    m->d_area_taps.reserve(m->area_taps.size() * sizeof(AreaTap));
    m->d_area_idx.reserve(m->area_idx.size() * 4);
    m->d_area_recs.reserve(std::max<size_t>(m->area_recs.size() * sizeof(AreaRec), 64));
    HIP_CHECK(hipMemcpyAsync(m->d_area_geoms.p, m->area_geoms.data(), m->area_geoms.size() * sizeof(AreaGeom), hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(m->d_area_taps.p, m->area_taps.data(), m->area_taps.size() * sizeof(AreaTap), hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(m->d_area_idx.p, m->area_idx.data(), m->area_idx.size() * 4, hipMemcpyHostToDevice, m->stream));
    if (!m->area_recs.empty()) HIP_CHECK(hipMemcpyAsync(m->d_area_recs.p, m->area_recs.data(), m->area_recs.size() * sizeof(AreaRec), hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));
    m->area_dirty = false;
}

// max frames of size (w,h) per unit under the workspace budget (the slots share it)
int sub_batch_for(slideo_matcher* m, const PyrGeom& g, int n) {
    size_t per = (size_t)g.frame_bytes * (blur_is_f32(m) ? 2 : 1) + (size_t)g.cand_per_frame * 4 + (size_t)g.nlevels * 258 * 4 + (size_t)g.w * g.h * 3;
    // downstream of ORB, sized by the per-frame keypoint capacity: items 8 + keypoint 24 + descriptor 32 B, the key lists
    // (32 x 4 B, times the train-set segments of a small query set: at most ~4 at sizes where the budget matters), and per
    // (keypoint, neighbour) the vote 8 B + point pair 16 B + mask 1 B
    const size_t kc = kp_cap_for(m, g);
    per += kc * (8 + sizeof(slideo_keypoint) + 32 + (size_t)KLIST * 4 * 4 + (size_t)m->cfg.knn_k * 25) + sizeof(FrameCands) + MAXR * sizeof(PairDesc);
    size_t fit = std::max<size_t>(1, (m->ws_budget / NSLOTS) / std::max<size_t>(per, 1));
    return (int)std::min<size_t>({(size_t)std::max(n, 1), fit, (size_t)4096});
}

void require_idle(slideo_matcher* m) {
    for (const Slot& S : m->slots) if (S.busy) fail(SLIDEO_ERR_STATE, "a submitted unit has not been collected yet");
}

uint8_t* stage_for_upload(slideo_matcher* m, size_t bytes) {
    m->kept.valid = false;             // slideo_match_kept_frames reads this buffer: whatever the mask call left there is overwritten
    m->slots[0].d_stage.reserve(bytes + 16);
    return m->slots[0].d_stage.as<uint8_t>();
}

// copies n host frames into S.d_stage with frame stride h*stride; `cs` != null: on that (copy) stream, and S.st waits for it
void upload_frames(Slot& S, const uint8_t* host, int n, int h, int stride, int64_t frame_stride, hipStream_t cs) {
    const size_t fb = (size_t)h * stride;
    S.d_stage.reserve(fb * n + 16);
    hipStream_t st = cs ? cs : S.st;
    if ((size_t)frame_stride == fb) {
        HIP_CHECK(hipMemcpyAsync(S.d_stage.p, host, fb * n, hipMemcpyHostToDevice, st));
    } else {
        for (int i = 0; i < n; ++i)
            HIP_CHECK(hipMemcpyAsync(S.d_stage.as<uint8_t>() + fb * i, host + (size_t)frame_stride * i, fb, hipMemcpyHostToDevice, st));
    }
    if (cs) {
        HIP_CHECK(hipEventRecord(S.ev_up, cs));
        HIP_CHECK(hipStreamWaitEvent(S.st, S.ev_up, 0));
    }
}

// page-locked (hipHostMalloc / hipHostRegister) host memory?  Copies from it are truly asynchronous DMA; copies from pageable
// memory are staged by the runtime inside the call.
bool host_is_pinned(const void* p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

void validate_image(int w, int h, int stride) {
    if (w < 1 || h < 1 || stride < w * 3) fail(SLIDEO_ERR_INVALID_ARG, "bad image geometry w=%d h=%d stride=%d", w, h, stride);
}

// ---- one unit of the per-frame hot path: enqueue everything, then collect ---------------
// `frames_dev` must stay valid until the unit is collected (reproject reads the frames).
// keypoints per frame the capacity-sized path provides for: twice the quota (ties at a level's retainBest threshold are kept, so
// no finite bound is safe; a frame beyond it is detected on the device and the unit re-run through the exact-size path)
// the cv::RNG((uint64)-1) stream RANSACPointSetRegistrator draws its samples from, pre-drawn (ptsetreg.cpp: rng state
// starts at -1 on every call, so every candidate reads the same stream from position 0)
void upload_rng_stream(slideo_matcher* m, uint32_t len) {
    std::vector<uint32_t> rng(len);
    CvRng r((uint64_t)-1, m->cfg.ocv.rng_mul);
    for (auto& v : rng) v = r.next();
    m->d_rng.reserve(rng.size() * 4);
    HIP_CHECK(hipMemcpy(m->d_rng.p, rng.data(), rng.size() * 4, hipMemcpyHostToDevice));
    m->rng_len = len;
}

uint32_t kp_cap_for(const slideo_matcher* m, const PyrGeom& g) {
    int cap = std::max(2 * m->cfg.nfeatures, m->cfg.nfeatures + 1024);
    cap = std::min(cap, KP_SORT_LDS);
    return (uint32_t)std::max(1, std::min(cap, std::max(g.cand_per_frame, 1)));
}

void unit_submit(slideo_matcher* m, Slot& S, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride,
                 bool allow_async) {
    // (does this unit share the chip with others?  the search then runs one block per CU: stage_knn.hip share_pad)
    { bool others = m->units_pending; for (const Slot& o : m->slots) others |= (&o != &S && o.busy); S.u_shared = others; }
    if (m->sift_on) { unit_submit_sift(m, S, frames_dev, n, w, h, stride, frame_stride); return; }
    const slideo_config& c = m->cfg;
    hipStream_t st = S.st;
    const bool prof = m->profiling;
    const PyrGeom& g = geom_for(m, w, h).g;
    // Capacity-sized (no host wait in the middle of the unit) when the matrix-core kNN runs: every kernel downstream of the ORB
    // counts reads them on the device.  The VALU engine (A/B only) keeps the exact-size path.
    const uint32_t kpcap = kp_cap_for(m, g);
    // (a capacity below quota + margin — the KP_SORT_LDS clamp at nfeatures >= ~7 k — would overflow on every busy frame and run
    // every unit twice: those configurations take the exact-size path from the start)
    const bool async = allow_async && m->async_submit && !knn_unit_is_valu(m, n * (int)std::min<uint32_t>(kpcap, (uint32_t)c.nfeatures)) &&
                       (int64_t)n * kpcap < ((int64_t)1 << 30) &&
                       (kpcap >= (uint32_t)c.nfeatures + 1024u || kpcap >= (uint32_t)std::max(g.cand_per_frame, 1));
    S.timed = prof; S.u_frames = frames_dev; S.u_w = w; S.u_h = h; S.u_stride = stride; S.u_fs = frame_stride; S.u_async = async;
    if (m->orb_chain && m->last_orb_ev && m->last_orb_ev != S.ev_orb) HIP_CHECK(hipStreamWaitEvent(st, m->last_orb_ev, 0));
    if (prof) HIP_CHECK(hipEventRecord(S.ev[0], st));
    orb_stage1(m, S, frames_dev, n, w, h, stride, frame_stride, false, async ? kpcap : 0xFFFFFFFFu);
    uint32_t qtot, qplan;
    if (async) {
        qtot = (uint32_t)n * kpcap;                                          // capacity
        qplan = (uint32_t)n * std::min<uint32_t>(kpcap, (uint32_t)c.nfeatures);   // what the kNN plan assumes
        S.orb.qtot = qtot; S.orb.max_count = kpcap;
    } else {
        orb_wait_info(m, S);                  // the other units' kNN / verify keep the GPU busy meanwhile
        qtot = qplan = S.orb.qtot;
    }
    // all workspace before the timed kNN interval
    S.u_nt = knn_unit_rows(m, (int)qplan);
    // the search's block shape while units share the chip (stage_knn.hip knn_shape): how much search there is per pixel of ORB work
    S.u_w12 = m->knn_w12_ratio > 0.0 && (double)qplan * (double)S.u_nt >= m->knn_w12_ratio * (double)n * (double)w * (double)h;
    knn_reserve_unit(m, S, qplan, qtot);
    S.d_votes.reserve(std::max<size_t>((size_t)qtot * c.knn_k * sizeof(uint2), 16));
    S.d_gpts.reserve(std::max<size_t>((size_t)qtot * c.knn_k * sizeof(float4), 16));
    S.d_gmask.reserve(std::max<size_t>((size_t)qtot * c.knn_k, 16));
    S.d_fcs.reserve((size_t)n * sizeof(FrameCands));
    S.d_verdicts.reserve((size_t)n * sizeof(slideo_verdict));
    S.d_pairs.reserve((size_t)n * MAXR * sizeof(PairDesc) + 64);
    S.h_out.reserve((size_t)n * (sizeof(slideo_verdict) + sizeof(FrameCands)) + 64);
    orb_stage2(m, S, w, h, async);
    HIP_CHECK(hipEventRecord(S.ev_orb, st));
    m->last_orb_ev = S.ev_orb;
    VerifyParams vp = make_vp(c);
    vp.rng_len = m->rng_len;
    HIP_CHECK(hipMemsetAsync(S.d_fcs.p, 0, (size_t)n * sizeof(FrameCands), st));
    if (prof) HIP_CHECK(hipEventRecord(S.ev[1], st));
    if (qtot > 0 && S.st_knn) {                                         // SLIDEO_CU_SPLIT: the search on its own CUs, ordered by events
        HIP_CHECK(hipEventRecord(S.ev_k0, st));
        HIP_CHECK(hipStreamWaitEvent(S.st_knn, S.ev_k0, 0));
        unit_knn(m, S, n, qplan, qtot, async, prof, S.st_knn);
        HIP_CHECK(hipEventRecord(S.ev_k1, S.st_knn));
        HIP_CHECK(hipStreamWaitEvent(st, S.ev_k1, 0));
    } else
    if (qtot > 0) unit_knn(m, S, n, qplan, qtot, async, prof, st);      // (records S.ev[2] behind the search when profiling)
    unit_verify(m, S, vp, frames_dev, n, w, h, stride, frame_stride, qtot);
}

void unit_collect(slideo_matcher* m, Slot& S, slideo_verdict* out_host) {
    const int n = S.n;
    HIP_CHECK(hipStreamSynchronize(S.st));
    S.busy = false;
    const uint8_t* ho = S.h_out.as<uint8_t>();
    const size_t tail = (size_t)n * (sizeof(slideo_verdict) + sizeof(FrameCands));
    uint32_t fl, info[2];
    std::memcpy(&fl, ho + tail, 4);
    std::memcpy(info, ho + tail + 4, 8);
    if (S.u_async) {
        if (fl & 1u) fail(SLIDEO_ERR_HIP, "internal: FAST candidate list overflow");
        if (fl & 8u) {
            // a frame had more keypoints than the capacity-sized path provides for (ties at a retainBest threshold are kept, as
            // in OpenCV): the whole unit again, through the exact-size path
            unit_submit(m, S, S.u_frames, n, S.u_w, S.u_h, S.u_stride, S.u_fs, false);
            unit_collect(m, S, out_host);
            return;
        }
        S.orb.qtot = info[0]; S.orb.max_count = info[1];
    }
    const uint32_t qtot = S.orb.qtot;
    if (fl & 4u) {
        // a candidate's sample schedule (2 draws per iteration + the redraws of equal pairs; 4 per attempt for the homography) ran
        // past the pre-drawn stream: draw four times as much and run the unit again.  (Other units may be reading the table:
        // drain the device first.)
        const uint32_t cap = 1u << 26;
        if (m->rng_len >= cap) fail(SLIDEO_ERR_CAPACITY, "RANSAC sample schedule exceeded %u pre-drawn RNG outputs", m->rng_len);
        HIP_CHECK(hipDeviceSynchronize());
        upload_rng_stream(m, (uint32_t)std::min<uint64_t>((uint64_t)m->rng_len * 4, cap));
        unit_submit(m, S, S.u_frames, n, S.u_w, S.u_h, S.u_stride, S.u_fs, false);
        unit_collect(m, S, out_host);
        return;
    }
    if (S.timed) {                    // (after both re-run checks: a unit that was run twice is counted once, by its final run)
        float t;
        HIP_CHECK(hipEventElapsedTime(&t, S.ev[0], S.ev[1])); m->prof_ms[0] += t; m->prof_n[0]++;
        if (qtot > 0) {
            HIP_CHECK(hipEventElapsedTime(&t, S.ev[1], S.ev[2])); m->prof_ms[1] += t; m->prof_n[1]++;
            m->prof_pairs += (int64_t)qtot * S.u_nt;       // pairs EVALUATED: unique train rows when the set is de-duplicated
            HIP_CHECK(hipEventElapsedTime(&t, S.ev[2], S.ev[3])); m->prof_ms[2] += t; m->prof_n[2]++;
        }
        HIP_CHECK(hipEventElapsedTime(&t, S.ev[0], S.ev[4])); m->prof_ms[3] += t; m->prof_n[3]++;
    }
    std::memcpy(out_host, ho, (size_t)n * sizeof(slideo_verdict));
    const size_t base = m->last_fcs.size();
    m->last_fcs.resize(base + n);
    std::memcpy(m->last_fcs.data() + base, ho + (size_t)n * sizeof(slideo_verdict), (size_t)n * sizeof(FrameCands));
}

void check_match_args(slideo_matcher* m, int n, const void* frames, const void* out, int w, int h, int stride, int64_t frame_stride) {
    if (!m->finalized) fail(SLIDEO_ERR_STATE, "slideo_matcher_finalize_pages must be called before matching");
    if (m->M <= 0) fail(SLIDEO_ERR_EMPTY_INDEX, "no page produced a descriptor");
    if (n < 0 || (n > 0 && (!frames || !out))) fail(SLIDEO_ERR_INVALID_ARG, "null frames/verdicts");
    validate_image(w, h, stride);
    if (m->sift_on) sift_check_cfg(&m->sift_cfg, w, h);          // (the doubled image's coordinates travel in 13 bits)
    if (frame_stride < (int64_t)h * stride) fail(SLIDEO_ERR_INVALID_ARG, "frame_stride smaller than one frame");
}

// Synchronous matching of n frames: cut into units and run them through the slots as a pipeline.
void match_frames_impl(slideo_matcher* m, int n, const uint8_t* frames, bool on_device, int w, int h, int stride,
                       int64_t frame_stride, slideo_verdict* out, hipStream_t user_stream) {
    check_match_args(m, n, frames, out, w, h, stride, frame_stride);
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    m->last_fcs.clear();
    if (n == 0) return;
    GeomEntry& ge = geom_for(m, w, h);
    area_class_for(m, w, h);
    upload_area(m);
    int unit = sub_batch_for(m, ge.g, n);
    if (n >= 128 && unit >= (n + 1) / 2) unit = (n + 1) / 2;      // two halves overlap ORB with kNN / verify
    // Host frames: the call is bound by the H2D copies (6.2 MB per 1080p frame: 256 frames = 29 ms at 55 GB/s against 14 ms of
    // kernels), so what matters is that the copy engines never wait: short units, each copied on its slot's stream while the
    // units before it compute — with two halves the second half's kernels start only when all of it has arrived.
    if (!on_device && m->host_unit > 0 && n >= 2 * m->host_unit) unit = std::min(unit, m->host_unit);
    struct Pending { Slot* S; int ofs; };
    std::vector<Pending> pend;
    if (on_device && user_stream)
        for (Slot& S : m->slots) {      // inputs produced on the caller's stream: order our streams behind it
            HIP_CHECK(hipEventRecord(S.ev_in, user_stream));
            HIP_CHECK(hipStreamWaitEvent(S.st, S.ev_in, 0));
        }
    int done = 0;
    const bool src_pinned = !on_device && host_is_pinned(frames);
    m->units_pending = n > unit;
    try {
        for (int i = 0; i < n; i += unit) {
            const int cnt = std::min(unit, n - i);
            if ((int)pend.size() == NSLOTS) {
                unit_collect(m, *pend[0].S, out + pend[0].ofs);
                done += pend[0].S->n;
                pend.erase(pend.begin());
                if (m->progress) m->progress(m->progress_user, (uint64_t)done, (uint64_t)n, "Processing frames...");
            }
            Slot& S = m->slots[m->next_slot];
            m->next_slot = (m->next_slot + 1) % NSLOTS;
            const uint8_t* dev;
            int64_t fs = frame_stride;
            if (on_device) dev = frames + (int64_t)i * frame_stride;
            else {
                // pinned source: asynchronous copies, kept in submission order on the one copy stream (see copy_st); pageable
                // source: the runtime stages the copy inside the call, on the unit's own stream (measured: 32.7 ms per 256 frames
                // that way against 54.8 through the copy stream)
                upload_frames(S, frames + (int64_t)i * frame_stride, cnt, h, stride, frame_stride, src_pinned ? m->copy_st : nullptr);
                dev = S.d_stage.as<uint8_t>(); fs = (int64_t)h * stride;
                m->kept.valid = false;                       // (slot 0's staging buffer may be overwritten)
            }
            unit_submit(m, S, dev, cnt, w, h, stride, fs);
            pend.push_back({&S, i});
        }
        for (Pending& p : pend) {
            unit_collect(m, *p.S, out + p.ofs);
            done += p.S->n;
            if (m->progress) m->progress(m->progress_user, (uint64_t)done, (uint64_t)n, "Processing frames...");
        }
        m->units_pending = false;
    } catch (...) {
        m->units_pending = false;
        (void)hipStreamSynchronize(m->copy_st);          // (DMA from the caller's pinned buffer may still be running)
        for (Slot& S : m->slots) { (void)hipStreamSynchronize(S.st); S.busy = false; }
        throw;
    }
}

// ProcessedImage::compute (mo/lib.rs:92-131) over n host pages: ORB (or SIFT) + to_small_image, runs of equally sized pages
// batched; the analysed pages, in order, into `out` (not appended to the matcher: the N-device group analyses a share of the
// deck on every device and appends the whole deck everywhere).  Progress: one report per page, (base + i + 1) of `total`.
void analyse_pages(slideo_matcher* m, int n_pages, const uint8_t* const* data, const int32_t* width, const int32_t* height, const int32_t* stride_bytes,
                   std::vector<HostPage>& out, uint64_t progress_base, uint64_t progress_total) {
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    hipStream_t st = S.st;
    int i = 0;
    while (i < n_pages) {
        // group a run of equally sized pages into one batch
        const int w = width[i], h = height[i], stride = stride_bytes[i];
        if (!data[i]) fail(SLIDEO_ERR_INVALID_ARG, "page %d is null", i);
        validate_image(w, h, stride);
        GeomEntry& ge = geom_for(m, w, h);
        int cap = std::min(sub_batch_for(m, ge.g, n_pages - i), 64), cnt = 1;
        while (cnt < cap && width[i + cnt] == w && height[i + cnt] == h && stride_bytes[i + cnt] == stride && data[i + cnt]) ++cnt;
        const size_t fb = (size_t)h * stride;
        stage_for_upload(m, fb * cnt);
        for (int j = 0; j < cnt; ++j)
            HIP_CHECK(hipMemcpyAsync(S.d_stage.as<uint8_t>() + fb * j, data[i + j], fb, hipMemcpyHostToDevice, st));
        const size_t dbytes = m->sift_on ? 128 : 32;                       // descriptor bytes per keypoint
        if (m->sift_on) add_pages_sift(m, S, cnt, w, h, stride, (int64_t)fb);     // -> S.d_kp / S.d_desc / S.orb.qofs, like run_orb
        else run_orb(m, S, S.d_stage.as<uint8_t>(), cnt, w, h, stride, (int64_t)fb, true);
        const uint32_t qtot = S.orb.qtot;
        std::vector<slideo_keypoint> kp(qtot);
        std::vector<uint8_t> desc((size_t)qtot * dbytes);
        if (qtot) {
            HIP_CHECK(hipMemcpyAsync(kp.data(), S.d_kp.p, (size_t)qtot * sizeof(slideo_keypoint), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipMemcpyAsync(desc.data(), S.d_desc.p, (size_t)qtot * dbytes, hipMemcpyDeviceToHost, st));
        }
        int sw = 0, sh = 0;
        run_small(m, S.d_stage.as<uint8_t>(), cnt, w, h, stride, (int64_t)fb, sw, sh, st);
        std::vector<uint8_t> smalls((size_t)cnt * sw * sh * 3);
        HIP_CHECK(hipMemcpyAsync(smalls.data(), m->d_small.p, smalls.size(), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        const int ac = area_class_for(m, w, h);
        for (int j = 0; j < cnt; ++j) {
            HostPage pg;
            pg.w = w; pg.h = h; pg.sw = sw; pg.sh = sh; pg.area_idx = ac;
            const uint32_t a = S.orb.qofs[j], b = S.orb.qofs[j + 1];
            pg.kp.assign(kp.begin() + a, kp.begin() + b);
            pg.desc.assign(desc.begin() + (size_t)a * dbytes, desc.begin() + (size_t)b * dbytes);
            pg.small_img.assign(smalls.begin() + (size_t)j * sw * sh * 3, smalls.begin() + (size_t)(j + 1) * sw * sh * 3);
            out.push_back(std::move(pg));
            if (m->progress) m->progress(m->progress_user, progress_base + (uint64_t)(i + j + 1), progress_total, "Analyzing PDF pages...");   // lib.rs:49-53
        }
        i += cnt;
    }
}

void append_page(slideo_matcher* m, const HostPage& pg) {
    HostPage c = pg;
    c.area_idx = area_class_for(m, pg.w, pg.h);          // (the size class index is per matcher)
    if (c.sw != m->area_geoms[c.area_idx].dw || c.sh != m->area_geoms[c.area_idx].dh)
        fail(SLIDEO_ERR_INVALID_ARG, "small image %dx%d is not the to_small_image size %dx%d of a %dx%d page", c.sw, c.sh,
             m->area_geoms[c.area_idx].dw, m->area_geoms[c.area_idx].dh, c.w, c.h);
    m->pages.push_back(std::move(c));
}

void set_err(slideo_matcher* m, const char* what) {
    if (m) m->err = what;
    else { std::lock_guard<std::mutex> lk(g_err_mutex); g_create_error = what; }
}

}  // namespace slideo

extern "C" {

uint32_t slideo_abi_version(void) { return SLIDEO_ABI_VERSION; }

void slideo_config_default(slideo_config* c) {
    if (!c) return;
    c->nfeatures = 2000; c->scale_factor = 1.2f; c->nlevels = 8; c->edge_threshold = 62;
    c->patch_size = 62; c->fast_threshold = 20;
    c->knn_k = 30; c->vote_tolerance = 1.05f; c->max_candidate_pages = 40;
    c->ransac_threshold = 3.0; c->ransac_max_iters = 2000; c->ransac_confidence = 0.99; c->refine_iters = 10;
    c->max_rated = 10; c->min_rating = 50.0; c->min_rating_ratio = 0.2;
    c->min_similarity = 0.5f; c->small_area = 300 * 400; c->changed_similarity = 0.98f;
    c->ratio_test = 0.0f;
    c->verify_model = 0;                             // the reference's estimateAffinePartial2D
    c->matcher = 0; c->lsh_tables = 6; c->lsh_key_bits = 12; c->lsh_multi_probe = 1;      // exact search; mo/flann.rs:16-18
    c->verdict_rule = 0;                             // mo/lib.rs:370-389: the best re-projection similarity wins
    std::memset(&c->ocv, 0, sizeof(c->ocv));         // every OpenCV-variant switch at its default
    c->ocv.rng_mul = 4164903690u;                    // CV_RNG_COEFF
    c->ocv.hdlt = 1;                                 // (verify_model 1 only, no reference counterpart: the form proven identical end to end, 1/60 of form 0's cost)
}

const char* slideo_last_error(const slideo_matcher* m) {
    if (m) return m->err.c_str();
    std::lock_guard<std::mutex> lk(g_err_mutex);
    static thread_local std::string copy;
    copy = g_create_error;
    return copy.c_str();
}

// ---- the slots' streams: on hardware queues of their own -------------------------------------------------------------------
// The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and two streams
// on one queue run their kernels one after the other.  The pipeline needs its NSLOTS slot streams to run BESIDE each other (the
// search of one unit over the ORB / verify kernels of the others): when the host has created streams before the matcher — an
// initialised RCCL communicator has — two slot streams would share a queue and the same job runs 10 % slower
// (profiles/r06_experiments.txt 6).  So the streams are picked by measurement: a candidate is kept iff a kernel on it completes
// while spin kernels keep every stream kept so far busy.  ~0.5 ms per candidate at create time; SLIDEO_STREAM_PICK=0: plain
// creation order.  (One 3-line kernel: this unit otherwise holds none.)
__global__ void slideo_spin_kernel(long long ticks) {           // ticks of the 100 MHz wall clock
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static void pick_independent_streams(hipStream_t* out, int n) {
    using clk = std::chrono::steady_clock;
    std::vector<hipStream_t> rejected;
    int have = 0;
    for (int attempt = 0; have < n && attempt < 4 * n + 8; ++attempt) {
        hipStream_t c = nullptr;
        HIP_CHECK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        slideo_spin_kernel<<<1, 64, 0, c>>>(0);                          // (its hardware queue comes into being with its first kernel)
        HIP_CHECK(hipStreamSynchronize(c));
        bool ok = have == 0;
        for (int trial = 0; !ok && trial < 2; ++trial) {                   // (twice: a host thread descheduled for 300 us must not cost a good stream)
            for (int i = 0; i < have; ++i) slideo_spin_kernel<<<1, 64, 0, out[i]>>>(60000);      // 600 us on every stream kept so far
            const auto t0 = clk::now();
            slideo_spin_kernel<<<1, 64, 0, c>>>(0);
            HIP_CHECK(hipStreamSynchronize(c));
            const double us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            for (int i = 0; i < have; ++i) HIP_CHECK(hipStreamSynchronize(out[i]));
            ok = us < 300.0;                                               // behind a spinning stream it would have waited the 600
        }
        if (ok) out[have++] = c; else rejected.push_back(c);
    }
    for (; have < n; ++have) {                                             // fewer independent queues than slots (GPU_MAX_HW_QUEUES < NSLOTS): whatever comes
        if (!rejected.empty()) { out[have] = rejected.back(); rejected.pop_back(); }
        else HIP_CHECK(hipStreamCreateWithFlags(&out[have], hipStreamNonBlocking));
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
}

int32_t slideo_matcher_create(const slideo_config* cfg, int32_t device, slideo_matcher** out) {
    slideo_matcher* m = nullptr;
    API_TRY
    if (!cfg || !out) fail(SLIDEO_ERR_INVALID_ARG, "null cfg/out");
    *out = nullptr;
    const char* why = "";
    if (!config_supported(*cfg, &why)) fail(SLIDEO_ERR_UNSUPPORTED, "unsupported config: %s", why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) fail(SLIDEO_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) fail(SLIDEO_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        fail(SLIDEO_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    HIP_CHECK(hipSetDevice(device));
    std::unique_ptr<slideo_matcher> mm(new slideo_matcher());
    mm->cfg = *cfg; mm->device = device;
    // environment switches of a matcher (include/slideo_amd.h, "Environment"); none changes a result
    { const long v = env_long("SLIDEO_KNN_ENGINE", 0); if (v >= 0 && v <= 3) mm->knn_engine = (int)v; }
    { const long v = env_long("SLIDEO_KNN_SHARE", -1); if ((v >= -1 && v <= 1) || (v >= 3 && v <= 6)) mm->knn_share = (int)v; }
    if (const char* e = std::getenv("SLIDEO_KNN_W12_RATIO")) mm->knn_w12_ratio = std::atof(e);
    mm->knn_nseg_force = (int)std::max(0l, std::min(64l, env_long("SLIDEO_KNN_NSEG", 0)));
    mm->async_submit = env_long("SLIDEO_ASYNC_SUBMIT", 1) != 0;
    mm->knn_dedup = env_long("SLIDEO_KNN_DEDUP", 1) != 0;
    if (const char* e = std::getenv("SLIDEO_LSH_ENGINE")) mm->lsh_gather = std::string(e) == "gather";
    mm->host_unit = (int)std::max(0l, env_long("SLIDEO_HOST_UNIT", 32));
    mm->orb_chain = env_long("SLIDEO_ORB_CHAIN", 1) != 0;
    {   // a third of a 288 GB device for the SIFT pyramids (a pass of 256 1080p frames: 90 GB; three passes of 86 at 24 GB cost 6 ms of 68);
        // an upper bound only — every call also stays under half of the memory that is free when it runs (stage_sift.hip sift_batch)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b >= ((size_t)192 << 30)) mm->sift_ws_mb = 96l << 10;
    }
    if (const char* e = std::getenv("SLIDEO_WS_GB")) { double gb = std::atof(e); if (gb > 0.1) mm->ws_budget = (size_t)(gb * (double)((size_t)1 << 30)); }
    // SLIDEO_CU_SPLIT=N (measurement only: VERDICT r04 asked for spatial instead of per-CU sharing): N of the 256 CUs for the
    // search streams, the rest for the slots' own streams.  Bit i of a HIP CU mask may be CU (i / 8) of XCD (i % 8) or CU (i % 32) of
    // XCD (i / 32) depending on the runtime; the pattern below enables the same number of CUs in every XCD under either reading.
    mm->cu_split = (int)std::min(248l, std::max(0l, env_long("SLIDEO_CU_SPLIT", 0))) / 8 * 8;
    const int cu_split_others = (int)env_long("SLIDEO_CU_SPLIT_OTHERS", 1);      // 0: only the search is confined, the other stages may run anywhere
    uint32_t mask_knn[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mask_rest[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (mm->cu_split) {
        for (int i = 0; i < 256; ++i) {
            const int a = i & 7, b = (i >> 3) & 3, c = i >> 5;
            const bool knn = ((a + c) & 7) * 4 + b < mm->cu_split / 8;
            (knn ? mask_knn : mask_rest)[i >> 5] |= 1u << (i & 31);
        }
        if (mm->knn_share < 0) mm->knn_share = 0;    // the search has its CUs to itself: two blocks per CU (SLIDEO_KNN_SHARE=1 still forces one)
    }
    hipStream_t picked[NSLOTS];
    const bool pick = !mm->cu_split && env_long("SLIDEO_STREAM_PICK", 1) != 0;
    if (pick) pick_independent_streams(picked, NSLOTS);
    int slot_i = 0;
    for (Slot& S : mm->slots) {
        if (pick) {
            S.st = picked[slot_i++];
            for (auto& e : S.ev) HIP_CHECK(hipEventCreate(&e));
            HIP_CHECK(hipEventCreateWithFlags(&S.ev_in, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&S.ev_orb, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&S.ev_up, hipEventDisableTiming));
            continue;
        }
        if (mm->cu_split) {
            // (hipExtStreamCreateWithCUMask has no flags argument: these are BLOCKING streams — they synchronise implicitly with the
            // legacy NULL stream, unlike the hipStreamNonBlocking streams of the normal path; a caller with default-stream work of its
            // own distorts what the switch measures)
            if (cu_split_others) HIP_CHECK(hipExtStreamCreateWithCUMask(&S.st, 8, mask_rest));
            else HIP_CHECK(hipStreamCreateWithFlags(&S.st, hipStreamNonBlocking));
            HIP_CHECK(hipExtStreamCreateWithCUMask(&S.st_knn, 8, mask_knn));
            HIP_CHECK(hipEventCreateWithFlags(&S.ev_k0, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&S.ev_k1, hipEventDisableTiming));
        } else
        HIP_CHECK(hipStreamCreateWithFlags(&S.st, hipStreamNonBlocking));
        for (auto& e : S.ev) HIP_CHECK(hipEventCreate(&e));
        HIP_CHECK(hipEventCreateWithFlags(&S.ev_in, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&S.ev_orb, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&S.ev_up, hipEventDisableTiming));
    }
    HIP_CHECK(hipStreamCreateWithFlags(&mm->copy_st, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&mm->sift_ev, hipEventDisableTiming));
    mm->stream = mm->slots[0].st;
    orb_stage_init(mm.get());
    {
        // similarity: 2 draws per iteration + redraws; homography: 4 per attempt, several attempts per accepted subset
        int64_t len = std::max<int64_t>(RNG_TABLE_MIN, (cfg->verify_model == 1 ? 64ll : 4ll) * std::max(cfg->ransac_max_iters, 1) + 2048);
        len = std::max<int64_t>(512, env_long("SLIDEO_RNG_STREAM_LEN", (long)len));                        // tests: force the growth path
        upload_rng_stream(mm.get(), (uint32_t)std::min<int64_t>(len, 1ll << 26));
    }
    verify_stage_init(mm.get());
    m = mm.release();
    *out = m;
    API_CATCH(nullptr)
}

#ifdef KT_PROBE
extern "C++" { namespace slideo { void knn_probe_report(); } }
#endif
void slideo_matcher_destroy(slideo_matcher* m) {
#ifdef KT_PROBE
    if (m) { (void)hipDeviceSynchronize(); slideo::knn_probe_report(); }
#endif
    if (!m) return;
    (void)hipSetDevice(m->device);
    for (Slot& S : m->slots) {
        if (S.st) { (void)hipStreamSynchronize(S.st); (void)hipStreamDestroy(S.st); }
        if (S.st_knn) { (void)hipStreamSynchronize(S.st_knn); (void)hipStreamDestroy(S.st_knn); }
        if (S.ev_k0) (void)hipEventDestroy(S.ev_k0);
        if (S.ev_k1) (void)hipEventDestroy(S.ev_k1);
        for (auto& e : S.ev) if (e) (void)hipEventDestroy(e);
        if (S.ev_in) (void)hipEventDestroy(S.ev_in);
        if (S.ev_orb) (void)hipEventDestroy(S.ev_orb);
        if (S.ev_up) (void)hipEventDestroy(S.ev_up);
    }
    if (m->copy_st) (void)hipStreamDestroy(m->copy_st);
    if (m->sift_ev) (void)hipEventDestroy(m->sift_ev);
    delete m;
}

int32_t slideo_matcher_max_in_flight(const slideo_matcher* m) { return m ? NSLOTS : 0; }

int32_t slideo_matcher_set_knn_engine(slideo_matcher* m, int32_t engine) {
    if (!m || engine < 0 || engine > 3) return SLIDEO_ERR_INVALID_ARG;
    m->knn_engine = engine;
    return SLIDEO_OK;
}

int32_t slideo_matcher_set_knn_exact_lists(slideo_matcher* m, int32_t on) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    m->knn_exact_lists = on ? 1 : 0;
    return SLIDEO_OK;
}

int32_t slideo_matcher_set_profiling(slideo_matcher* m, int32_t enable) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    HIP_CHECK(hipSetDevice(m->device));
    m->profiling = enable != 0;
    for (int i = 0; i < SLIDEO_N_STAGES; ++i) { m->prof_ms[i] = 0; m->prof_n[i] = 0; }
    m->prof_pairs = 0;
    m->d_clk.reserve(64);
    HIP_CHECK(hipMemset(m->d_clk.p, 0, 64));
    API_CATCH(m)
}

int32_t slideo_matcher_read_shader_clock(slideo_matcher* m, double* mhz_out, int64_t* samples_out) {
    if (!m || !mhz_out) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    *mhz_out = 0.0;
    if (samples_out) *samples_out = 0;
    if (!m->d_clk.p) return SLIDEO_OK;                       // profiling was never on
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    unsigned long long v[3] = {0, 0, 0};
    HIP_CHECK(hipMemcpy(v, m->d_clk.p, sizeof(v), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemset(m->d_clk.p, 0, 64));
    if (v[1] > 0) *mhz_out = 100.0 * (double)v[0] / (double)v[1];      // s_memrealtime counts 100 MHz
    if (samples_out) *samples_out = (int64_t)v[2];
    API_CATCH(m)
}

int32_t slideo_matcher_read_profile(slideo_matcher* m, double* ms_out, int64_t* launches_out, int64_t* knn_pairs_out) {
    if (!m || !ms_out || !launches_out) return SLIDEO_ERR_INVALID_ARG;
    for (int i = 0; i < SLIDEO_N_STAGES; ++i) { ms_out[i] = m->prof_ms[i]; launches_out[i] = m->prof_n[i]; m->prof_ms[i] = 0; m->prof_n[i] = 0; }
    if (knn_pairs_out) *knn_pairs_out = m->prof_pairs;
    m->prof_pairs = 0;
    return SLIDEO_OK;
}

int32_t slideo_matcher_set_progress(slideo_matcher* m, slideo_progress_fn fn, void* user) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    m->progress = fn; m->progress_user = user;
    return SLIDEO_OK;
}

int32_t slideo_matcher_add_pages_bgr8(slideo_matcher* m, int32_t n_pages, const uint8_t* const* data, const int32_t* width,
                                      const int32_t* height, const int32_t* stride_bytes) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (m->finalized) fail(SLIDEO_ERR_STATE, "pages cannot be added after finalize");
    if (n_pages < 0 || (n_pages > 0 && (!data || !width || !height || !stride_bytes))) fail(SLIDEO_ERR_INVALID_ARG, "null page arrays");
    const uint64_t total = (uint64_t)n_pages;
    if (m->progress) m->progress(m->progress_user, 0, total, "Analyzing PDF pages...");     // lib.rs:43
    std::vector<HostPage> got;
    analyse_pages(m, n_pages, data, width, height, stride_bytes, got, 0, total);
    for (const HostPage& pg : got) append_page(m, pg);
    if (m->progress) m->progress(m->progress_user, total, total, "PDF page analysis successful.");   // lib.rs:58
    API_CATCH(m)
}

int32_t slideo_matcher_add_page_features(slideo_matcher* m, int32_t width, int32_t height, int32_t n_keypoints, const slideo_keypoint* kp,
                                         const uint8_t* desc32, const uint8_t* small_bgr, int32_t small_w, int32_t small_h) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (m->finalized) fail(SLIDEO_ERR_STATE, "pages cannot be added after finalize");
    if (m->sift_on) fail(SLIDEO_ERR_UNSUPPORTED, "page features are 32-byte ORB descriptors: not in SIFT mode");
    if (n_keypoints < 0 || (n_keypoints > 0 && (!kp || !desc32)) || !small_bgr) fail(SLIDEO_ERR_INVALID_ARG, "null page feature arrays");
    validate_image(width, height, width * 3);
    HIP_CHECK(hipSetDevice(m->device));
    const int ac = area_class_for(m, width, height);              // (also checks that the page is large enough for to_small_image)
    if (small_w != m->area_geoms[ac].dw || small_h != m->area_geoms[ac].dh)
        fail(SLIDEO_ERR_INVALID_ARG, "small image %dx%d is not the to_small_image size %dx%d of a %dx%d page", small_w, small_h,
             m->area_geoms[ac].dw, m->area_geoms[ac].dh, width, height);
    HostPage pg;
    pg.w = width; pg.h = height; pg.sw = small_w; pg.sh = small_h; pg.area_idx = ac;
    pg.kp.assign(kp, kp + n_keypoints);
    pg.desc.assign(desc32, desc32 + (size_t)n_keypoints * 32);
    pg.small_img.assign(small_bgr, small_bgr + (size_t)small_w * small_h * 3);
    m->pages.push_back(std::move(pg));
    API_CATCH(m)
}

int32_t slideo_matcher_get_page_small(const slideo_matcher* cm, int32_t page_idx, uint8_t* out, int64_t out_capacity, int32_t* sw, int32_t* sh) {
    slideo_matcher* m = const_cast<slideo_matcher*>(cm);
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (page_idx < 0 || page_idx >= (int)m->pages.size() || !sw || !sh) fail(SLIDEO_ERR_INVALID_ARG, "page %d out of range", page_idx);
    const HostPage& pg = m->pages[page_idx];
    *sw = pg.sw; *sh = pg.sh;
    if ((int64_t)pg.small_img.size() > out_capacity) fail(SLIDEO_ERR_CAPACITY, "small image needs %zu bytes", pg.small_img.size());
    if (out) std::memcpy(out, pg.small_img.data(), pg.small_img.size());
    API_CATCH(m)
}

int32_t slideo_matcher_finalize_pages(slideo_matcher* m) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (m->finalized) fail(SLIDEO_ERR_STATE, "already finalized");
    HIP_CHECK(hipSetDevice(m->device));
    const int P = (int)m->pages.size();
    if (P > 16384) fail(SLIDEO_ERR_UNSUPPORTED, "%d pages exceed the 16384 the vote kernel's LDS layout holds", P);
    int64_t M = 0, small_bytes = 0;
    for (const HostPage& p : m->pages) { M += (int64_t)p.kp.size(); small_bytes += (int64_t)p.small_img.size(); }
    if (M >= ((int64_t)1 << KNN_KEY_SHIFT)) fail(SLIDEO_ERR_UNSUPPORTED, "%lld descriptors exceed 2^23", (long long)M);
    // (the matcher stays open for more pages: the reference's FLANN train on an empty set throws, mo/flann.rs:45-47)
    if (M == 0) fail(SLIDEO_ERR_EMPTY_INDEX, "no page produced a descriptor");
    const size_t dbytes = m->sift_on ? 128 : 32;                           // descriptor bytes per row
    std::vector<uint8_t> train((size_t)M * dbytes);
    std::vector<int32_t> tpage((size_t)M);
    std::vector<float2> xy((size_t)M);
    std::vector<PageInfo> info(P);
    std::vector<uint8_t> smalls((size_t)small_bytes);
    int64_t row = 0, sofs = 0;
    for (int p = 0; p < P; ++p) {
        const HostPage& pg = m->pages[p];
        PageInfo& pi = info[p];
        pi.w = pg.w; pi.h = pg.h; pi.area_idx = pg.area_idx; pi.sw = pg.sw; pi.sh = pg.sh;
        pi.kp_ofs = (int32_t)row; pi.kp_cnt = (int32_t)pg.kp.size(); pi._pad = 0; pi.small_ofs = sofs;
        std::memcpy(train.data() + (size_t)row * dbytes, pg.desc.data(), pg.desc.size());
        for (size_t i = 0; i < pg.kp.size(); ++i) { tpage[row + i] = p; xy[row + i] = make_float2(pg.kp[i].x, pg.kp[i].y); }
        std::memcpy(smalls.data() + sofs, pg.small_img.data(), pg.small_img.size());
        row += (int64_t)pg.kp.size(); sofs += (int64_t)pg.small_img.size();
    }
    m->d_train_page.reserve(std::max<size_t>(tpage.size() * 4, 16));
    m->d_page_xy.reserve(std::max<size_t>(xy.size() * sizeof(float2), 16));
    m->d_pageinfo.reserve(std::max<size_t>(info.size() * sizeof(PageInfo), 16));
    m->d_page_small.reserve(smalls.size() + 16);       // (+ slack: reproject_vt_kernel reads a 3-byte pixel as one dword)
    if (M > 0 && m->sift_on) {
        // SIFT mode: the rows become the train set of the squared-L2 engine (norm order, centred tile-major operand)
        HIP_CHECK(hipMemcpy(m->d_train_page.p, tpage.data(), tpage.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(m->d_page_xy.p, xy.data(), xy.size() * sizeof(float2), hipMemcpyHostToDevice));
        l2_prepare(m->l2, train.data(), (int)M, m->stream);
        m->Mu = M;
    } else if (M > 0) {
        HIP_CHECK(hipMemcpy(m->d_train_page.p, tpage.data(), tpage.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(m->d_page_xy.p, xy.data(), xy.size() * sizeof(float2), hipMemcpyHostToDevice));
        knn_build_index(m, train, M);                  // the Hamming index: distinct rows -> matrix-core operand (+ LSH tables)
    }
    if (P > 0) {
        HIP_CHECK(hipMemcpy(m->d_pageinfo.p, info.data(), info.size() * sizeof(PageInfo), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(m->d_page_small.p, smalls.data(), smalls.size(), hipMemcpyHostToDevice));
    }
    upload_area(m);
    m->M = M;
    m->finalized = true;
    API_CATCH(m)
}

int32_t slideo_matcher_page_count(const slideo_matcher* m) { return m ? (int32_t)m->pages.size() : -1; }
int64_t slideo_matcher_descriptor_count(const slideo_matcher* m) { return m && m->finalized ? m->M : -1; }
int64_t slideo_matcher_unique_descriptor_count(const slideo_matcher* m) { return m && m->finalized ? m->Mu : -1; }

int32_t slideo_matcher_get_page_features(const slideo_matcher* cm, int32_t page_idx, slideo_keypoint* kp, uint8_t* desc32,
                                         int32_t capacity, int32_t* n_out) {
    slideo_matcher* m = const_cast<slideo_matcher*>(cm);
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (page_idx < 0 || page_idx >= (int)m->pages.size()) fail(SLIDEO_ERR_INVALID_ARG, "page %d out of range", page_idx);
    const HostPage& pg = m->pages[page_idx];
    if (n_out) *n_out = (int32_t)pg.kp.size();
    if ((int)pg.kp.size() > capacity) fail(SLIDEO_ERR_CAPACITY, "page has %zu keypoints, capacity %d", pg.kp.size(), capacity);
    if (kp) std::memcpy(kp, pg.kp.data(), pg.kp.size() * sizeof(slideo_keypoint));
    if (desc32) std::memcpy(desc32, pg.desc.data(), pg.desc.size());
    API_CATCH(m)
}

int32_t slideo_match_frames_bgr8(slideo_matcher* m, int32_t n_frames, const uint8_t* frames, int32_t width, int32_t height,
                                 int32_t stride_bytes, int64_t frame_stride_bytes, slideo_verdict* verdicts_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    match_frames_impl(m, n_frames, frames, false, width, height, stride_bytes, frame_stride_bytes, verdicts_out, nullptr);
    API_CATCH(m)
}

int32_t slideo_match_frames_bgr8_dev(slideo_matcher* m, int32_t n_frames, const uint8_t* frames_dev, int32_t width, int32_t height,
                                     int32_t stride_bytes, int64_t frame_stride_bytes, slideo_verdict* verdicts_out, void* hip_stream) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    match_frames_impl(m, n_frames, frames_dev, true, width, height, stride_bytes, frame_stride_bytes, verdicts_out,
                      reinterpret_cast<hipStream_t>(hip_stream));
    API_CATCH(m)
}

int32_t slideo_match_frames_submit_dev(slideo_matcher* m, int32_t n_frames, const uint8_t* frames_dev, int32_t width, int32_t height,
                                       int32_t stride_bytes, int64_t frame_stride_bytes, void* hip_stream, int64_t* ticket_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (!ticket_out) fail(SLIDEO_ERR_INVALID_ARG, "null ticket_out");
    check_match_args(m, n_frames, frames_dev, ticket_out, width, height, stride_bytes, frame_stride_bytes);
    if (n_frames < 1) fail(SLIDEO_ERR_INVALID_ARG, "submit needs at least one frame");
    HIP_CHECK(hipSetDevice(m->device));
    Slot& S = m->slots[m->next_slot];
    if (S.busy) fail(SLIDEO_ERR_STATE, "all slots are in flight: collect ticket %lld first", (long long)S.ticket);
    GeomEntry& ge = geom_for(m, width, height);
    if (n_frames > sub_batch_for(m, ge.g, n_frames))
        fail(SLIDEO_ERR_CAPACITY, "%d frames exceed the per-slot workspace budget (%d); submit smaller units or raise SLIDEO_WS_GB",
             n_frames, sub_batch_for(m, ge.g, n_frames));
    area_class_for(m, width, height);
    upload_area(m);
    { bool any = false; for (const Slot& c : m->slots) any |= c.busy; if (!any) m->last_fcs.clear(); }
    if (hip_stream) {
        HIP_CHECK(hipEventRecord(S.ev_in, reinterpret_cast<hipStream_t>(hip_stream)));
        HIP_CHECK(hipStreamWaitEvent(S.st, S.ev_in, 0));
    }
    unit_submit(m, S, frames_dev, n_frames, width, height, stride_bytes, frame_stride_bytes);
    S.ticket = m->next_ticket++;
    *ticket_out = S.ticket;
    m->next_slot = (m->next_slot + 1) % NSLOTS;
    for (Slot& O : m->slots) if (&O != &S && !O.busy) O.match_capacity(S);  // the next units find their workspace sized
    API_CATCH(m)
}

int32_t slideo_match_frames_collect_dev(slideo_matcher* m, int64_t ticket, slideo_verdict* verdicts_out, void* verdicts_dev_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (!verdicts_out) fail(SLIDEO_ERR_INVALID_ARG, "null verdicts_out");
    HIP_CHECK(hipSetDevice(m->device));
    Slot* S = nullptr;
    for (Slot& c : m->slots) if (c.busy && c.ticket == ticket) S = &c;
    if (!S) fail(SLIDEO_ERR_STATE, "ticket %lld is not in flight", (long long)ticket);
    for (Slot& c : m->slots) if (c.busy && c.ticket < ticket) fail(SLIDEO_ERR_STATE, "collect ticket %lld first (in order)", (long long)c.ticket);
    unit_collect(m, *S, verdicts_out);
    if (verdicts_dev_out) {           // (after the collect: a unit re-run through the exact-size path has rewritten d_verdicts)
        HIP_CHECK(hipMemcpyAsync(verdicts_dev_out, S->d_verdicts.p, (size_t)S->n * sizeof(slideo_verdict), hipMemcpyDeviceToDevice, S->st));
        HIP_CHECK(hipStreamSynchronize(S->st));
    }
    API_CATCH(m)
}

int32_t slideo_match_frames_collect(slideo_matcher* m, int64_t ticket, slideo_verdict* verdicts_out) {
    return slideo_match_frames_collect_dev(m, ticket, verdicts_out, nullptr);
}

int32_t slideo_last_frame_candidates(const slideo_matcher* m, int32_t frame_in_batch, slideo_candidate* out, int32_t capacity,
                                     int32_t* n_out) {
    if (!m || !n_out) return SLIDEO_ERR_INVALID_ARG;
    if (frame_in_batch < 0 || frame_in_batch >= (int)m->last_fcs.size()) return SLIDEO_ERR_INVALID_ARG;
    const FrameCands& fc = m->last_fcs[frame_in_batch];
    *n_out = fc.ncand;
    if (fc.ncand > capacity) return SLIDEO_ERR_CAPACITY;
    for (int i = 0; i < fc.ncand; ++i) {
        slideo_candidate& c = out[i];
        c.page_idx = fc.page[i]; c.n_votes = fc.count[i]; c.inliers = fc.inliers[i]; c.survived = 0; c.similarity = 0.f;
        if (m->cfg.verify_model == 1) { for (int j = 0; j < 9; ++j) c.transform[j] = fc.M[i][j]; }
        else {
            for (int j = 0; j < 6; ++j) c.transform[j] = fc.M[i][j];
            c.transform[6] = c.transform[7] = 0.0; c.transform[8] = fc.found[i] ? 1.0 : 0.0;
        }
        for (int s = 0; s < fc.nsurv; ++s) if (fc.surv[s] == i) { c.survived = 1; c.similarity = fc.sim[s]; }
    }
    return SLIDEO_OK;
}

int32_t slideo_changed_mask_bgr8(slideo_matcher* m, int32_t n_frames, const uint8_t* frames, int32_t width, int32_t height,
                                 int32_t stride_bytes, int64_t frame_stride_bytes, const uint8_t* prev_small,
                                 uint8_t* last_small_out, uint8_t* changed_out, float* similarity_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (n_frames < 0 || (n_frames > 0 && (!frames || !changed_out))) fail(SLIDEO_ERR_INVALID_ARG, "null frames/changed");
    validate_image(width, height, stride_bytes);
    if (n_frames == 0) return SLIDEO_OK;
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    hipStream_t st = S.st;
    int sw = 0, sh = 0;
    const size_t fb = (size_t)height * stride_bytes;
    stage_for_upload(m, fb * (size_t)n_frames);
    upload_frames(S, frames, n_frames, height, stride_bytes, frame_stride_bytes);
    m->kept = slideo_matcher::Kept{true, n_frames, width, height, stride_bytes};   // stays in slot 0's staging buffer: slideo_match_kept_frames
    run_small(m, S.d_stage.as<uint8_t>(), n_frames, width, height, stride_bytes, (int64_t)fb, sw, sh, st);
    const size_t sb = (size_t)sw * sh * 3;
    DevBuf& prev = m->d_prev_small;
    prev.reserve(sb);
    if (prev_small) HIP_CHECK(hipMemcpyAsync(prev.p, prev_small, sb, hipMemcpyHostToDevice, st));
    m->d_ssd.reserve((size_t)n_frames * 8);
    // pair i: (small[i-1], small[i]); pair 0 uses prev
    if (prev_small) launch_ssd(prev.as<uint8_t>(), 0, m->d_small.as<uint8_t>(), 0, (int64_t)sb, m->d_ssd.as<unsigned long long>(), 1, st);
    if (n_frames > 1)
        launch_ssd(m->d_small.as<uint8_t>(), (int64_t)sb, m->d_small.as<uint8_t>() + sb, (int64_t)sb, (int64_t)sb, m->d_ssd.as<unsigned long long>() + 1, n_frames - 1, st);
    std::vector<unsigned long long> ssd(n_frames, 0);
    HIP_CHECK(hipMemcpyAsync(ssd.data(), m->d_ssd.p, (size_t)n_frames * 8, hipMemcpyDeviceToHost, st));
    if (last_small_out)
        HIP_CHECK(hipMemcpyAsync(last_small_out, m->d_small.as<uint8_t>() + sb * (n_frames - 1), sb, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    for (int i = 0; i < n_frames; ++i) {
        float sim = 0.0f;   // video_capture.rs:92: the first frame compares as 0.0
        if (i > 0 || prev_small) {
            double e = std::sqrt((double)ssd[i]);
            float max_error = std::sqrt((255.0f * 255.0f * 3.0f) * (float)(sw * sh));
            sim = 1.0f - (float)e / max_error;
        }
        changed_out[i] = sim < m->cfg.changed_similarity ? 1 : 0;
        if (similarity_out) similarity_out[i] = sim;
    }
    API_CATCH(m)
}

int32_t slideo_match_kept_frames(slideo_matcher* m, int32_t n_sel, const int32_t* sel, slideo_verdict* verdicts_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (n_sel < 0 || (n_sel > 0 && (!sel || !verdicts_out))) fail(SLIDEO_ERR_INVALID_ARG, "null selection/verdicts");
    if (!m->kept.valid) fail(SLIDEO_ERR_STATE, "no frames kept: slideo_changed_mask_bgr8 must be the call before (its upload is what is matched)");
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    const slideo_matcher::Kept k = m->kept;
    const size_t fb = (size_t)k.h * k.stride;
    for (int i = 0; i < n_sel; ++i) if (sel[i] < 0 || sel[i] >= k.n) fail(SLIDEO_ERR_INVALID_ARG, "selected frame %d outside the %d kept", sel[i], k.n);
    if (n_sel == 0) return SLIDEO_OK;
    // the selected frames packed back to back (device to device: 6 MB per 1080p frame at HBM speed), runs of consecutive
    // indices in one copy
    m->d_kept.reserve(fb * (size_t)n_sel + 16);
    hipStream_t st = m->slots[0].st;
    for (int i = 0; i < n_sel;) {
        int j = i + 1;
        while (j < n_sel && sel[j] == sel[j - 1] + 1) ++j;
        HIP_CHECK(hipMemcpyAsync(m->d_kept.as<uint8_t>() + fb * i, m->slots[0].d_stage.as<uint8_t>() + fb * sel[i], fb * (size_t)(j - i), hipMemcpyDeviceToDevice, st));
        i = j;
    }
    match_frames_impl(m, n_sel, m->d_kept.as<uint8_t>(), true, k.w, k.h, k.stride, (int64_t)fb, verdicts_out, st);
    API_CATCH(m)
}

// Pins a caller's frame buffer (hipHostRegister) so that the H2D copies of slideo_match_frames_bgr8 / slideo_changed_mask_bgr8
// read it by DMA without the runtime's staging copy.  Worth it for a buffer that is reused across calls (a decoder's frame ring):
// registering costs about as much as one copy of the buffer.
int32_t slideo_host_register(void* ptr, size_t bytes) {
    if (!ptr || !bytes) return SLIDEO_ERR_INVALID_ARG;
    return hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess ? SLIDEO_OK : SLIDEO_ERR_HIP;
}
int32_t slideo_host_unregister(void* ptr) {
    if (!ptr) return SLIDEO_ERR_INVALID_ARG;
    return hipHostUnregister(ptr) == hipSuccess ? SLIDEO_OK : SLIDEO_ERR_HIP;
}

}  // extern "C"
