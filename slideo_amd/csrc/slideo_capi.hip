// slideo_capi.hip — implementation of include/slideo_amd.h for gfx950.
//
// Host-side runtime of the matcher: device-resident page database, per-frame-size
// geometry cache, growing workspace, batched kernel pipeline on one HIP stream.
// No CPU fallback exists: every compute entry point needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "geom.h"
#include "knn.hip.h"
#include "knn_tile.hip.h"
#include "knn_l2.hip.h"
#include "knn_lsh.hip.h"
#include "orb.hip.h"
#include "slideo_amd.h"
#include "verify.hip.h"
#include "homography.hip.h"
#include "sift.hip.h"

using namespace slideo;

namespace {

#ifndef SLIDEO_NSLOTS
#define SLIDEO_NSLOTS 4
#endif
constexpr int NSLOTS = SLIDEO_NSLOTS;      // units in flight (each with its own workspace and HIP stream)
constexpr int KLIST = 32;
static_assert(KLIST == VOTE_KLIST, "vote_kernel reads whole key lists");

struct GeomEntry {
    int w = 0, h = 0;
    PyrGeom g;
    DevBuf lin_tab;
    DevBuf fast_tiles;                // per FAST tile: level, origin, raw-column alignment (orb.hip.h fast_tile_entry)
    std::vector<PyrChain> chains;     // fused pyramid launches (empty: the per-level kernels)
    DevBuf spans;
    size_t chain_lds_max = 0;
};

struct HostPage {
    int w = 0, h = 0, sw = 0, sh = 0, area_idx = -1;
    std::vector<slideo_keypoint> kp;
    std::vector<uint8_t> desc;
    std::vector<uint8_t> small_img;
};

struct OrbOut {            // where the last ORB run of a slot left its results (device)
    uint32_t qtot = 0, max_count = 0;
    int nframes = 0;
    bool full_blur = false;       // stage 1 materialised the whole blurred pyramid (pyramid tap)
    std::vector<uint32_t> qofs;   // host copy, nframes+1
};

// One workspace + stream.  Two slots let the ORB stage of one unit of frames run concurrently with the
// kNN / verification stages of the previous unit (matrix-core bound vs VALU/LDS/HBM bound work).
struct Slot {
    hipStream_t st = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_in = nullptr, ev_orb = nullptr, ev_up = nullptr, ev_knn = nullptr;
    // arguments of the unit in flight (re-run through the exact-size path if the capacity-sized one overflowed)
    const uint8_t* u_frames = nullptr; int u_w = 0, u_h = 0, u_stride = 0; int64_t u_fs = 0; bool u_async = false; int u_nt = 0;
    DevBuf d_stage, d_pyr, d_blur, d_cand, d_hist, d_candcount, d_flags, d_thr, d_lvlofs, d_kpcount, d_qofs, d_info;
    DevBuf d_items, d_kp, d_desc, d_keys, d_knn_pend, d_votes, d_gpts, d_gmask, d_fcs, d_verdicts, d_pairs, d_blurmask, d_qkeys, d_tail, d_refine;
    PinBuf h_info, h_out;
    OrbOut orb;
    // unit in flight
    bool busy = false;
    int64_t ticket = 0;
    int n = 0;
    bool timed = false;

    // Give this (idle) slot the capacities of `o`.  A slot used for the first time would otherwise grow its ~25 buffers
    // (hipFree + hipMalloc, device-wide stalls, tens of ms for the GB-sized ones) in the middle of a steady-state
    // stream of batches; sizing every idle slot when one grows keeps every later unit allocation-free.
    void match_capacity(const Slot& o) {
        DevBuf* mine[] = {&d_stage, &d_pyr, &d_blur, &d_cand, &d_hist, &d_candcount, &d_flags, &d_thr, &d_lvlofs, &d_kpcount, &d_qofs,
                          &d_info, &d_items, &d_kp, &d_desc, &d_keys, &d_knn_pend, &d_votes, &d_gpts, &d_gmask, &d_fcs, &d_verdicts, &d_pairs, &d_blurmask};
        const DevBuf* theirs[] = {&o.d_stage, &o.d_pyr, &o.d_blur, &o.d_cand, &o.d_hist, &o.d_candcount, &o.d_flags, &o.d_thr, &o.d_lvlofs,
                                  &o.d_kpcount, &o.d_qofs, &o.d_info, &o.d_items, &o.d_kp, &o.d_desc, &o.d_keys, &o.d_knn_pend, &o.d_votes,
                                  &o.d_gpts, &o.d_gmask, &o.d_fcs, &o.d_verdicts, &o.d_pairs, &o.d_blurmask};
        static_assert(sizeof(mine) / sizeof(mine[0]) == sizeof(theirs) / sizeof(theirs[0]), "same buffer lists");
        for (size_t i = 0; i < sizeof(mine) / sizeof(mine[0]); ++i) mine[i]->reserve_cap(theirs[i]->cap);
        h_info.reserve_cap(o.h_info.cap); h_out.reserve_cap(o.h_out.cap);
    }
};

}  // namespace

struct slideo_matcher {
    slideo_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;      // = slots[0].st (setup, page ingest, taps)
    std::string err;
    slideo_progress_fn progress = nullptr;
    void* progress_user = nullptr;
    size_t ws_budget = (size_t)48 << 30;      // all slots together (SLIDEO_WS_GB); 288 GB of HBM per GPU

    DevBuf d_tables, d_rng, d_ictab;
    struct L2Set { DevBuf d_tx, d_tn, d_side, d_perm, d_keys, d_pend; int nt = 0, nt_pad = 0; bool ready = false; } l2;   // cfg2: the L2 train set
    uint32_t rng_len = 0;
    int ic_shift = 0, ic_entries = 0;     // intensity-centroid weight table of describe_kernel (geom.h ic_weight_table)
    std::vector<std::unique_ptr<GeomEntry>> geoms;

    // INTER_AREA size classes
    std::vector<AreaGeom> area_geoms;
    std::vector<AreaTap> area_taps;
    std::vector<int32_t> area_idx;
    DevBuf d_area_geoms, d_area_taps, d_area_idx;
    bool area_dirty = true;

    // pages
    std::vector<HostPage> pages;
    bool finalized = false;
    // slideo_matcher_use_sift: SIFT features + L2 k-NN (k = 2) + ratio test in front of the verify stage.  The SIFT and L2
    // workspaces are the matcher's (not a slot's): the extraction stages of consecutive units take turns (sift_ev)
    bool sift_on = false;
    slideo_sift_config sift_cfg{};
    float sift_ratio = 0.75f;
    hipEvent_t sift_ev = nullptr;
    bool sift_ev_set = false;
    int64_t M = -1;
    DevBuf d_train, d_train_page, d_page_xy, d_pageinfo, d_page_small;
    DevBuf d_trainb, d_train_side, d_train_nminh, d_train_perm;   // {0,1} FP4 operand in norm order + its side arrays (knn_tile.hip.h)
    // train-set de-duplication (knn.hip.h knn_expand_dups_kernel): the matrix-core engine searches the Mu unique rows, keys carry
    // the lowest original row of a group, d_grp_next chains the equal rows.  SLIDEO_KNN_DEDUP=0 searches all M rows.
    DevBuf d_utrain, d_grp_next;
    struct LshSet { DevBuf ofs, rows, keys; LshDev dev{}; bool ready = false; } lsh;      // slideo_config.matcher 1 (knn_lsh.hip.h)
    int64_t Mu = -1;
    int knn_dedup = 1;
    int lsh_gather = 0;              // SLIDEO_LSH_ENGINE=gather: matcher 1 through knn_lsh_kernel (buckets gathered) instead of the filtered matrix-core stream
    int host_unit = 32;              // frames per unit of a HOST-memory batch (SLIDEO_HOST_UNIT; 0 = the device-path rule)
    // every H2D copy of frame units goes through ONE stream, in submission order: copies issued on the units' own streams run
    // concurrently and share the link, so the first unit's frames arrive when all of them have (measured: 39 - 45 ms per 256
    // frames from pinned memory against 31 in order)
    hipStream_t copy_st = nullptr;
    // the frames slideo_changed_mask_bgr8 uploaded last (slot 0's staging buffer), for slideo_match_kept_frames
    struct Kept { bool valid = false; int n = 0, w = 0, h = 0, stride = 0; } kept;
    DevBuf d_kept;
    int knn_engine = 0;     // 0 = FP4 MFMA, wave shape chosen per launch (default), 1 = integer VALU popcount,
                            // 2 = FP4 MFMA, 2 waves/SIMD x 4 query tiles (knn_tile4_kernel), 3 = 4 waves/SIMD x 2 tiles (knn_tile2_kernel)
    int knn_exact_lists = 0;  // 1 = the matcher's kNN stage keeps full exact k-NN lists (no fused vote filter)
    // Units are enqueued in one go, without the mid-unit host wait for the keypoint counts: everything downstream of the ORB
    // counts is sized by capacity and reads the counts on the device (SLIDEO_ASYNC_SUBMIT=0: the exact-size path with the wait).
    int async_submit = 1;
    // ORB stages of consecutive units take turns (each waits for the previous unit's ORB stage on the GPU, event to event):
    // what the host wait used to enforce as a side effect (SLIDEO_ORB_CHAIN=0: free-running).
    int orb_chain = 1;
    int pyr_chain = 0;               // SLIDEO_PYR_CHAIN=1: fused pyramid launches (pyr_chain_kernel) instead of gray_kernel + one resize_kernel per level.
                                     // Off: measured SLOWER — 1.24 + 0.69 + 0.2 ms for the three launches against 0.41 + 1.2 ms
    hipEvent_t last_orb_ev = nullptr;
    // The search kernels of consecutive units take turns too (SLIDEO_KNN_CHAIN): two launches that share the CUs each last twice
    // as long for the same throughput
    int knn_chain = 0;
    hipEvent_t last_knn_ev = nullptr;

    // workspaces
    Slot slots[NSLOTS];
    int next_slot = 0;
    int64_t next_ticket = 1;
    DevBuf d_small, d_ssd, d_prev_small, d_tapq, d_tapt, d_tapidx, d_tapdist;
    struct SiftWs { DevBuf base, gauss, gray, cand, counts, raw, items, kept, qofs, info, kp, desc; } sift;     // csrc/sift.hip.h

    // stage profiling (HIP events on the launch streams)
    bool profiling = false;
    double prof_ms[SLIDEO_N_STAGES] = {0, 0, 0, 0};
    int64_t prof_n[SLIDEO_N_STAGES] = {0, 0, 0, 0};
    int64_t prof_pairs = 0;

    // trace of the last match call
    std::vector<FrameCands> last_fcs;
};

namespace {

std::string g_create_error;
std::mutex g_err_mutex;

VerifyParams make_vp(const slideo_config& c) {
    VerifyParams v{};
    v.k = c.knn_k; v.klist = KLIST; v.max_cand = c.max_candidate_pages; v.max_rated = c.max_rated;
    v.tol = c.vote_tolerance; v.min_similarity = c.min_similarity; v.ratio = c.ratio_test;
    v.thr = c.ransac_threshold; v.conf = c.ransac_confidence; v.min_rating = c.min_rating;
    v.min_rating_ratio = c.min_rating_ratio; v.max_iters = c.ransac_max_iters; v.refine_iters = c.refine_iters;
    v.model = c.verify_model;
    const char* e = getenv("SLIDEO_RANSAC_WINDOW");      // (read per unit: the tests switch it)
    v.sched_window = e ? (atoi(e) != 0) : 1;
    return v;
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
    if (tab.empty()) tab.push_back(0);
    e->lin_tab.reserve(tab.size() * 4);
    HIP_CHECK(hipMemcpyAsync(e->lin_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, m->stream));
    {
        std::vector<int4> ft((size_t)std::max(e->g.fast_tiles, 1));
        for (int t = 0; t < e->g.fast_tiles; ++t) ft[t] = fast_tile_entry(e->g, t);
        e->fast_tiles.reserve(ft.size() * sizeof(int4));
        HIP_CHECK(hipMemcpyAsync(e->fast_tiles.p, ft.data(), ft.size() * sizeof(int4), hipMemcpyHostToDevice, m->stream));
        HIP_CHECK(hipStreamSynchronize(m->stream));
    }
    // fused pyramid chains: the LDS form of the resize step reads a group's taps from three dwords (shrink factor <= 2)
    std::vector<PyrSpan> spans;
    if (m->pyr_chain && m->cfg.scale_factor <= 2.0f) build_pyr_chains(e->g, tab, e->chains, spans);
    if (!e->chains.empty()) {
        e->spans.reserve(std::max<size_t>(spans.size() * sizeof(PyrSpan), 16));
        HIP_CHECK(hipMemcpyAsync(e->spans.p, spans.data(), spans.size() * sizeof(PyrSpan), hipMemcpyHostToDevice, m->stream));
        for (const PyrChain& c : e->chains)
            e->chain_lds_max = std::max(e->chain_lds_max, (size_t)c.buf_bytes[0] + c.buf_bytes[1] + (size_t)c.xt_entries * 8 + (size_t)c.yt_entries * 4);
        if (e->chain_lds_max > 60 * 1024) e->chains.clear();              // (never with 256 x 32 tiles; the per-level kernels then)
    }
    HIP_CHECK(hipStreamSynchronize(m->stream));
    m->geoms.push_back(std::move(e));
    return *m->geoms.back();
}

int area_class_for(slideo_matcher* m, int w, int h) {
    for (size_t i = 0; i < m->area_geoms.size(); ++i)
        if (m->area_geoms[i].sw == w && m->area_geoms[i].sh == h) return (int)i;
    AreaGeom a;
    if (!build_area_geom(w, h, m->cfg.small_area, a, m->area_taps, m->area_idx, m->cfg.ocv.area))
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
    HIP_CHECK(hipMemcpyAsync(m->d_area_geoms.p, m->area_geoms.data(), m->area_geoms.size() * sizeof(AreaGeom), hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(m->d_area_taps.p, m->area_taps.data(), m->area_taps.size() * sizeof(AreaTap), hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(m->d_area_idx.p, m->area_idx.data(), m->area_idx.size() * 4, hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));
    m->area_dirty = false;
}

bool blur_is_f32(const slideo_matcher* m) { return m->cfg.ocv.blur <= 1; }

// max frames of size (w,h) per unit under the workspace budget (the slots share it)
uint32_t kp_cap_for(const slideo_matcher* m, const PyrGeom& g);
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

// ---- ORB over `n` equally sized frames already on the device, in three steps -------------
// stage 1: gray, pyramid, FAST+NMS, blur, retainBest thresholds, per-frame offsets; copies {Qtot, max, flags} to pinned memory
void launch_blur(slideo_matcher* m, Slot& S, const PyrGeom& g, int n, const uint8_t* strip_mask) {
    hipStream_t st = S.st;
    if (m->cfg.ocv.blur == 0)
        blur_f32_kernel<true><<<dim3(g.blur_tiles, n), 256, 0, st>>>(g, S.d_pyr.as<uint8_t>(), S.d_blur.as<uint8_t>(), m->d_tables.as<OrbTables>(), strip_mask);
    else if (m->cfg.ocv.blur == 1)
        blur_f32_kernel<false><<<dim3(g.blur_tiles, n), 256, 0, st>>>(g, S.d_pyr.as<uint8_t>(), S.d_blur.as<uint8_t>(), m->d_tables.as<OrbTables>(), strip_mask);
    else
        blur_kernel<<<dim3(g.blur_tiles, n), 256, 0, st>>>(g, S.d_pyr.as<uint8_t>(), S.d_blur.as<uint8_t>(), m->d_tables.as<OrbTables>());
    check_launch("blur kernel");
}

// `with_blur`: also materialise the WHOLE blurred pyramid (only the pyramid tap wants it)
// the f32 blur of ocv.blur 0 / 1 cannot be evaluated per BRIEF sample in integer arithmetic: those variants always
// materialise the blurred pyramid (blur_f32_kernel) and describe from it (describe_blurred_kernel)
void orb_stage1(slideo_matcher* m, Slot& S, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride,
                bool with_blur = false, uint32_t kp_cap = 0xFFFFFFFFu) {
    hipStream_t st = S.st;
    const bool full_blur = with_blur;
    with_blur = with_blur || blur_is_f32(m);
    GeomEntry& ge = geom_for(m, w, h);
    const PyrGeom& g = ge.g;
    const int L = g.nlevels;
    S.d_pyr.reserve((size_t)g.frame_bytes * n + 256);      // + slack: describe_kernel stages whole dwords past a window's last byte
    if (with_blur) S.d_blur.reserve((size_t)g.frame_bytes * n);
    S.d_cand.reserve(std::max<size_t>((size_t)g.cand_per_frame * n * 4, 16));
    const size_t n_cc = (size_t)n * L;
    S.d_hist.reserve(n_cc * 256 * 4);
    S.d_candcount.reserve(n_cc * 2 * 4);
    S.d_flags.reserve(16);
    uint32_t* hist = S.d_hist.as<uint32_t>();
    uint32_t* cand_count = S.d_candcount.as<uint32_t>();
    uint32_t* flags = S.d_flags.as<uint32_t>();
    HIP_CHECK(hipMemsetAsync(hist, 0, n_cc * 256 * 4, st));
    HIP_CHECK(hipMemsetAsync(cand_count, 0, n_cc * 2 * 4, st));
    HIP_CHECK(hipMemsetAsync(flags, 0, 16, st));
    S.d_thr.reserve(n_cc * 4); S.d_lvlofs.reserve(n_cc * 4); S.d_kpcount.reserve((size_t)n * 4);
    S.d_qofs.reserve((size_t)(n + 1) * 4); S.d_info.reserve(64);
    S.h_info.reserve(64);

    const int aligned4 = ((uintptr_t)frames_dev % 4 == 0) && (stride % 4 == 0) && (frame_stride % 4 == 0);
    if (!ge.chains.empty()) {
        const GrayCoef gc = m->cfg.ocv.gray == 1 ? GrayCoef{1868u, 9617u, 4899u, 14u} : GrayCoef{3735u, 19235u, 9798u, 15u};
        for (const PyrChain& c : ge.chains) {
            const size_t lds = (size_t)c.buf_bytes[0] + c.buf_bytes[1] + (size_t)c.xt_entries * 8 + (size_t)c.yt_entries * 4;
            const dim3 grid(c.tiles_x * c.tiles_y, n);
            if (c.from_bgr)
                pyr_chain_kernel<true><<<grid, 256, lds, st>>>(g, c, ge.spans.as<PyrSpan>(), ge.lin_tab.as<uint32_t>(), frames_dev, frame_stride, stride,
                                                               aligned4, gc, S.d_pyr.as<uint8_t>());
            else
                pyr_chain_kernel<false><<<grid, 256, lds, st>>>(g, c, ge.spans.as<PyrSpan>(), ge.lin_tab.as<uint32_t>(), frames_dev, frame_stride, stride,
                                                                aligned4, gc, S.d_pyr.as<uint8_t>());
            check_launch("pyr_chain_kernel");
        }
    } else {
    {
        dim3 grid(cdiv(cdiv(w, 4), 256), h, n);
        const GrayCoef gc = m->cfg.ocv.gray == 1 ? GrayCoef{1868u, 9617u, 4899u, 14u} : GrayCoef{3735u, 19235u, 9798u, 15u};
        gray_kernel<<<grid, 256, 0, st>>>(frames_dev, frame_stride, stride, S.d_pyr.as<uint8_t>(), g.frame_bytes, w, h,
                                          g.lv[0].pitch, aligned4, gc);
        check_launch("gray_kernel");
    }
    for (int l = 1; l < L; ++l) {
        if (g.lv[l].w <= 0 || g.lv[l].h <= 0) continue;
        // flat thread index t -> (row, 4-pixel group) = (t / nxq, t % nxq); magic = ceil(2^32 / nxq) divides
        // exactly while nxq^2 * h < 2^32, which MAX_DIM guarantees
        const int nxq = cdiv(g.lv[l].w, 4);
        const uint32_t magic = nxq > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)nxq - 1) / (uint64_t)nxq) : 0u;
        dim3 grid(cdiv(nxq * cdiv(g.lv[l].h, RESIZE_ROWS), 256), 1, n);
        resize_kernel<<<grid, 256, 0, st>>>(S.d_pyr.as<uint8_t>(), g.frame_bytes, g.lv[l - 1], g.lv[l], ge.lin_tab.as<uint32_t>(),
                                            nxq, magic);
        check_launch("resize_kernel");
    }
    }
    if (g.fast_tiles > 0) {
        fast_kernel<<<dim3(cdiv(g.fast_tiles, FAST_TPB), n), 256, 0, st>>>(g, S.d_pyr.as<uint8_t>(), S.d_cand.as<uint32_t>(), cand_count, hist, ge.fast_tiles.as<int4>());
        check_launch("fast_kernel");
    }
    // the whole blurred pyramid only for the pyramid tap; on the frame path the f32 variants blur in stage 2, and only the strips
    // the kept keypoints sample (blur_mark_kernel)
    if (full_blur && g.blur_tiles > 0) launch_blur(m, S, g, n, nullptr);
    threshold_kernel<<<n, 64 * L, 0, st>>>(g, hist, cand_count, S.d_thr.as<uint32_t>(), S.d_lvlofs.as<uint32_t>(),
                                           S.d_kpcount.as<uint32_t>(), flags, kp_cap);
    check_launch("threshold_kernel");
    scan_kernel<<<1, 1024, 0, st>>>(S.d_kpcount.as<uint32_t>(), n, S.d_qofs.as<uint32_t>(), S.d_info.as<uint32_t>());
    check_launch("scan_kernel");
    HIP_CHECK(hipMemcpyAsync(S.h_info.p, S.d_info.p, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(S.h_info.as<uint32_t>() + 2, flags, 4, hipMemcpyDeviceToHost, st));
    S.orb.nframes = n;
    S.orb.full_blur = full_blur;
}

// the one mid-pipeline host sync: 12 bytes that size everything downstream
void orb_wait_info(slideo_matcher* m, Slot& S) {
    HIP_CHECK(hipStreamSynchronize(S.st));
    const uint32_t qtot = S.h_info.as<uint32_t>()[0], maxc = S.h_info.as<uint32_t>()[1], fl = S.h_info.as<uint32_t>()[2];
    if (fl & 1u) fail(SLIDEO_ERR_HIP, "internal: FAST candidate list overflow");
    S.orb.qtot = qtot; S.orb.max_count = maxc;
}

// stage 2: compact the kept candidates, canonical sort, IC angle + rotated BRIEF
// by_capacity: qtot / maxc are CAPACITIES (n * kp_cap, kp_cap) and the real counts stay on the device
void orb_stage2(slideo_matcher* m, Slot& S, int w, int h, bool by_capacity = false) {
    hipStream_t st = S.st;
    const PyrGeom& g = geom_for(m, w, h).g;
    const int L = g.nlevels, n = S.orb.nframes;
    const uint32_t qtot = S.orb.qtot, maxc = S.orb.max_count;
    const uint32_t qtot_arg = by_capacity ? 0xFFFFFFFFu : qtot;
    S.d_items.reserve(std::max<size_t>((size_t)qtot * 8, 16));
    S.d_kp.reserve(std::max<size_t>((size_t)qtot * sizeof(slideo_keypoint), 16));
    S.d_desc.reserve(std::max<size_t>((size_t)qtot * 32, 32));
    if (qtot == 0) return;
    uint32_t* cand_count = S.d_candcount.as<uint32_t>();
    uint32_t* cursor = cand_count + (size_t)n * L;
    compact_kernel<<<dim3(L, n), 256, 0, st>>>(g, S.d_cand.as<uint32_t>(), cand_count, S.d_thr.as<uint32_t>(),
                                               S.d_lvlofs.as<uint32_t>(), S.d_qofs.as<uint32_t>(), cursor, S.d_items.as<uint64_t>());
    check_launch("compact_kernel");
    int np2 = 2;
    while ((uint32_t)np2 < maxc && np2 < KP_SORT_LDS) np2 <<= 1;
    sort_kernel<<<n, 1024, (size_t)np2 * 8, st>>>(S.d_qofs.as<uint32_t>(), S.d_items.as<uint64_t>(), np2);
    check_launch("sort_kernel");
    if (maxc > (uint32_t)np2) {                       // only reachable through the exact-size path (kp_cap_for <= KP_SORT_LDS)
        sort_global_kernel<<<n, 1024, 0, st>>>(S.d_qofs.as<uint32_t>(), S.d_items.as<uint64_t>(), (uint32_t)np2);
        check_launch("sort_global_kernel");
    }
    if (blur_is_f32(m)) {
        if (!S.orb.full_blur && g.blur_tiles > 0) {
            const size_t mask_bytes = (size_t)n * g.blur_tiles * 4;
            S.d_blurmask.reserve(mask_bytes);
            HIP_CHECK(hipMemsetAsync(S.d_blurmask.p, 0, mask_bytes, st));
            blur_mark_kernel<<<cdiv((int)qtot, 256), 256, 0, st>>>(g, S.d_qofs.as<uint32_t>(), n, S.d_items.as<uint64_t>(), qtot_arg, S.d_blurmask.as<uint8_t>());
            check_launch("blur_mark_kernel");
            launch_blur(m, S, g, n, S.d_blurmask.as<uint8_t>());
        }
        describe_blurred_kernel<<<cdiv((int)qtot, 4), 256, 0, st>>>(g, S.d_pyr.as<uint8_t>(), S.d_blur.as<uint8_t>(), m->d_tables.as<OrbTables>(),
                                                                    S.d_qofs.as<uint32_t>(), n, S.d_items.as<uint64_t>(), qtot_arg,
                                                                    m->d_ictab.as<uint2>(), m->ic_shift, m->ic_entries,
                                                                    S.d_kp.as<slideo_keypoint>(), S.d_desc.as<uint8_t>(), m->cfg.ocv.atan);
        check_launch("describe_blurred_kernel");
        return;
    }
    const DescWin dw = describe_window(g.half_patch);
    describe_kernel<<<cdiv((int)qtot, 4), 256, (size_t)dw.dwords * 16, st>>>(g, S.d_pyr.as<uint8_t>(), m->d_tables.as<OrbTables>(),
                                                                             S.d_qofs.as<uint32_t>(), n, S.d_items.as<uint64_t>(), qtot_arg, dw,
                                                                             m->d_ictab.as<uint2>(), m->ic_shift, m->ic_entries,
                                                                             S.d_kp.as<slideo_keypoint>(), S.d_desc.as<uint8_t>(), m->cfg.ocv.atan);
    check_launch("describe_kernel");
}

// synchronous ORB (page ingest, taps).  Leaves: d_qofs[n+1], d_kp[qtot], d_desc[qtot*32]; S.orb filled.
void run_orb(slideo_matcher* m, Slot& S, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride,
             bool keep_host_qofs, bool with_blur = false) {
    orb_stage1(m, S, frames_dev, n, w, h, stride, frame_stride, with_blur);
    orb_wait_info(m, S);
    orb_stage2(m, S, w, h);
    if (keep_host_qofs) {
        S.orb.qofs.resize(n + 1);
        HIP_CHECK(hipMemcpyAsync(S.orb.qofs.data(), S.d_qofs.p, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost, S.st));
        HIP_CHECK(hipStreamSynchronize(S.st));
    }
}

// ---- exact Hamming kNN: keys into S.d_keys[0 .. nq*KLIST) ------------------------------
int knn_pad_rows(int nt) { return cdiv(std::max(nt, 1), KT_ST_ROWS) * KT_ST_ROWS; }

// Operand of the {0,1} x {0,1} engine (knn_tile.hip.h): rows in ascending popcount order (stable counting sort on the host:
// nt x 32 bytes of popcounts), expanded to tile-major FP4 on the device, plus per super-tile the rows' norms and original
// indices and per tile half its smallest norm.  `t_host`: the packed rows in host memory.
struct TrainBits { DevBuf *tx, *side, *nminh, *perm; };
// rowid (may be null): the row number a key carries for row i of t_host / t_dev (de-duplicated sets: the lowest original row)
void prepare_train_bits(const uint8_t* t_host, const uint32_t* t_dev, int nt, TrainBits o, hipStream_t st, const int32_t* rowid = nullptr) {
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
void build_lsh_set(const slideo_config& c, const uint8_t* t_host, int nt, slideo_matcher::LshSet& S, hipStream_t st) {
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

struct KnnPlan { int engine, qblocks, nseg, per_seg; };
// Engine 0 ("mfma") = the 2-tile wave shape (knn_tile2_kernel: 4 waves/SIMD, two 512-query blocks per CU) at every size: since the
// {0,1} operand alphabet it runs the headline launch in 10.0 ms alone against 11.3 for the 4-tile shape and the step is 2 %
// shorter.  The 4-tile shape (engine 2) and the VALU popcount kernel (engine 1) stay selectable for A/B: identical results.
int knn_engine_for(const slideo_matcher* m, int nq) {
    if (m->knn_engine != 0) return m->knn_engine;
    (void)nq;
    return 3;
}
// nq: the query count the plan is made for (the real one, or its estimate when only the device knows it); nq_grid >= nq:
// what the grid and the buffers are sized for (blocks past the device-side count leave at once)
KnnPlan knn_plan(const slideo_matcher* m, int nq, int nt, int nq_grid = 0) {
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
        p.qblocks = cdiv(nq, knn_qpb<2>());
        const int n_st = knn_pad_rows(nt) / KT_ST_ROWS;
        // the chip holds 512 blocks (two per CU).  From 3/4 of that on, one pass over the train set is best (every
        // segment pays its own list warm-up and the merge); fewer query blocks split the train set so that the blocks
        // fill the chip in ONE round (floor, not ceil: 1.4 rounds of smaller blocks lose more to the tail than the
        // empty slots do).  Measured (r01): 236 query blocks x 1.8 M rows (64 4K frames): 1 segment 33.2 ms, 2 segments 24.2 ms,
        // 3 segments 23.5 ms; 239 query blocks x 517 k rows (128 1080p frames): 2 segments 6.24 ms, 3 segments 6.65 ms
        int nseg = p.qblocks >= 384 ? 1 : std::min(std::max(512 / std::max(p.qblocks, 1), 1), n_st);
        static const int nseg_env = [] { const char* e = getenv("SLIDEO_KNN_NSEG"); return e ? atoi(e) : 0; }();   // (experiments)
        if (nseg_env > 0) nseg = std::min(nseg_env, n_st);
        p.per_seg = cdiv(n_st, std::max(nseg, 1));
        p.nseg = cdiv(n_st, p.per_seg);
        p.qblocks = cdiv(nq_grid, knn_qpb<2>());
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

void knn_reserve(slideo_matcher* m, Slot& S, int nq, int nt, int nq_grid = 0) {
    const KnnPlan p = knn_plan(m, nq, nt, nq_grid);
    S.d_keys.reserve((size_t)p.nseg * std::max(std::max(nq, nq_grid), 1) * KLIST * 4);
    if (p.engine == 2) S.d_knn_pend.reserve((size_t)p.qblocks * p.nseg * KT_WAVES * knn_pend_words_per_wave<4>() * 4);
    if (p.engine == 3) S.d_knn_pend.reserve((size_t)p.qblocks * p.nseg * KT_WAVES * knn_pend_words_per_wave<2>() * 4);
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
void run_knn(slideo_matcher* m, Slot& S, const uint32_t* q_dev, int nq, const TrainOps& T, int nt, float prune_tol,
             const uint32_t* nq_dev = nullptr, int nq_grid = 0) {
    if (nq <= 0 && !nq_dev) return;
    hipStream_t st = S.st;
    if ((int64_t)nt >= ((int64_t)1 << KNN_KEY_SHIFT)) fail(SLIDEO_ERR_UNSUPPORTED, "train set of %d rows exceeds %d", nt, 1 << KNN_KEY_SHIFT);
    const KnnPlan p = knn_plan(m, nq, nt, nq_grid);
    knn_reserve(m, S, nq, nt, nq_grid);
    const int nq_all = std::max(std::max(nq, nq_grid), 1);
    if ((p.engine == 2 || p.engine == 3) && nt > 0) {
        if (p.engine == 2)
            knn_tile4_kernel<<<dim3(p.qblocks, p.nseg), KT_THREADS, 0, st>>>(q_dev, nq, T.txb, T.side, T.nminh, knn_pad_rows(nt), p.per_seg,
                                                                             S.d_keys.as<uint32_t>(), S.d_knn_pend.as<uint32_t>(), prune_tol, nq_dev);
        else
            knn_tile2_kernel<<<dim3(p.qblocks, p.nseg), KT_THREADS, 0, st>>>(q_dev, nq, T.txb, T.side, T.nminh, knn_pad_rows(nt), p.per_seg,
                                                                             S.d_keys.as<uint32_t>(), S.d_knn_pend.as<uint32_t>(), prune_tol, nq_dev);
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

// ---- to_small_image of n equally sized device images into m->d_small --------------------
void run_small(slideo_matcher* m, const uint8_t* imgs_dev, int n, int w, int h, int stride, int64_t img_stride,
               int& sw, int& sh, hipStream_t st) {
    int ac = area_class_for(m, w, h);
    upload_area(m);
    const AreaGeom& ag = m->area_geoms[ac];
    sw = ag.dw; sh = ag.dh;
    m->d_small.reserve((size_t)n * sw * sh * 3);
    int tiles = cdiv(sw, SM_TW) * cdiv(sh, SM_TH);
    small_image_kernel<<<dim3(tiles, n), 256, 0, st>>>(ag, m->d_area_taps.as<AreaTap>(), m->d_area_idx.as<int32_t>(), imgs_dev,
                                                       img_stride, stride, m->d_small.as<uint8_t>(), (int64_t)sw * sh * 3);
    check_launch("small_image_kernel");
}

// copies n host frames into S.d_stage with frame stride h*stride; `cs` != null: on that (copy) stream, and S.st waits for it
void upload_frames(Slot& S, const uint8_t* host, int n, int h, int stride, int64_t frame_stride, hipStream_t cs = nullptr) {
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

// Everything after the neighbour lists (S.d_keys, Hamming key format) of a unit: the per-page vote, RANSAC (similarity or
// homography), rating, re-projection, verdicts, and the unit's one D2H copy.  `frames_dev`: the unit's frames (re-projection).
void unit_verify(slideo_matcher* m, Slot& S, const VerifyParams& vp, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride,
                 uint32_t qtot) {
    const slideo_config& c = m->cfg;
    hipStream_t st = S.st;
    const bool prof = m->profiling;
    const int P = (int)m->pages.size();
    uint32_t* flags = S.d_flags.as<uint32_t>();
    if (qtot > 0) {
        const size_t lds = (size_t)P * 4 + (((size_t)P + 15) & ~(size_t)15) + (size_t)c.max_candidate_pages * 256 * 4;
        vote_kernel<<<n, 256, lds, st>>>(vp, S.d_keys.as<uint32_t>(), S.d_qofs.as<uint32_t>(), m->d_train_page.as<int32_t>(), P,
                                         S.d_fcs.as<FrameCands>(), S.d_votes.as<uint2>());
        check_launch("vote_kernel");
        if (c.verify_model == 1) {
            // hdlt >= 1: a candidate still sampling after `round_cap` rounds goes to ransac_h_tail_kernel (8 waves on its sample
            // schedule).  hdlt 0 is bound by its eigen-solver, not by the schedule: no cap.  flags[1] counts the tail list, flags[2] hands it out (flags[3]: refine_h's eigenproblem list).
            const uint32_t tail_rounds = [] {                                 // (read per unit: the tests switch it)
                const char* e = getenv("SLIDEO_RH_TAIL_ROUNDS");
                const long v = e ? atol(e) : 256;
                return (uint32_t)(v < 1 ? 0xFFFFFFFFu : v);                   // 0 / negative: never hand over
            }();
            const uint32_t round_cap = c.ocv.hdlt ? tail_rounds : 0xFFFFFFFFu;
            S.d_tail.reserve((size_t)c.max_candidate_pages * n * 4 + 16);
            auto launch_h = [&](auto small_tag, auto large_tag, int hdlt) {
                small_tag<<<dim3(c.max_candidate_pages, n), 64, ransac_h_lds_bytes(RANSAC_SMALL_PTS, hdlt), st>>>(
                    vp, S.d_qofs.as<uint32_t>(), S.d_kp.as<slideo_keypoint>(), m->d_page_xy.as<float2>(), S.d_votes.as<uint2>(),
                    m->d_rng.as<uint32_t>(), S.d_fcs.as<FrameCands>(), S.d_gpts.as<float4>(), S.d_gmask.as<uint8_t>(), flags, round_cap,
                    S.d_tail.as<uint32_t>(), flags + 1);
                check_launch("ransac_h_kernel (small)");
                large_tag<<<dim3(c.max_candidate_pages, n), 64, ransac_h_lds_bytes(RANSAC_LDS_PTS, hdlt), st>>>(
                    vp, S.d_qofs.as<uint32_t>(), S.d_kp.as<slideo_keypoint>(), m->d_page_xy.as<float2>(), S.d_votes.as<uint2>(),
                    m->d_rng.as<uint32_t>(), S.d_fcs.as<FrameCands>(), S.d_gpts.as<float4>(), S.d_gmask.as<uint8_t>(), flags, round_cap,
                    S.d_tail.as<uint32_t>(), flags + 1);
                check_launch("ransac_h_kernel (large)");
            };
            auto launch_tail = [&](auto tail_tag) {
                const int blocks = std::min(RANSAC_H_TAIL_BLOCKS, c.max_candidate_pages * n);
                tail_tag<<<blocks, 64 * RANSAC_H_TAIL_WAVES, ransac_h_tail_lds_bytes(RANSAC_H_TAIL_WAVES), st>>>(
                    vp, S.d_qofs.as<uint32_t>(), S.d_kp.as<slideo_keypoint>(), m->d_page_xy.as<float2>(), S.d_votes.as<uint2>(),
                    m->d_rng.as<uint32_t>(), S.d_fcs.as<FrameCands>(), S.d_gpts.as<float4>(), S.d_gmask.as<uint8_t>(), flags,
                    S.d_tail.as<uint32_t>(), flags + 1, flags + 2);
                check_launch("ransac_h_tail_kernel");
            };
            if (c.ocv.hdlt == 2) {
                launch_h(&ransac_h_kernel<RANSAC_SMALL_PTS, 0, 2>, &ransac_h_kernel<RANSAC_LDS_PTS, RANSAC_SMALL_PTS + 1, 2>, 2);
                if (round_cap != 0xFFFFFFFFu) launch_tail(&ransac_h_tail_kernel<2, RANSAC_H_TAIL_WAVES>);
            } else if (c.ocv.hdlt == 1) {
                launch_h(&ransac_h_kernel<RANSAC_SMALL_PTS, 0, 1>, &ransac_h_kernel<RANSAC_LDS_PTS, RANSAC_SMALL_PTS + 1, 1>, 1);
                if (round_cap != 0xFFFFFFFFu) launch_tail(&ransac_h_tail_kernel<1, RANSAC_H_TAIL_WAVES>);
            } else launch_h(&ransac_h_kernel<RANSAC_SMALL_PTS, 0, 0>, &ransac_h_kernel<RANSAC_LDS_PTS, RANSAC_SMALL_PTS + 1, 0>, 0);
            if (c.refine_iters > 0) {
                // runKernel over the inliers: normalise + L^T L (wave per candidate), the eigenproblems 32 per wave, then the LM
                const size_t ncand = (size_t)c.max_candidate_pages * n;
                S.d_refine.reserve(ncand * sizeof(RefineRec) + ncand * 4 + 16);
                RefineRec* recs = S.d_refine.as<RefineRec>();
                uint32_t* eig_list = reinterpret_cast<uint32_t*>(S.d_refine.as<uint8_t>() + ncand * sizeof(RefineRec));
                refine_h_kernel<0><<<dim3(c.max_candidate_pages, n), 64, 0, st>>>(
                    vp, S.d_qofs.as<uint32_t>(), S.d_kp.as<slideo_keypoint>(), m->d_page_xy.as<float2>(), S.d_votes.as<uint2>(),
                    S.d_fcs.as<FrameCands>(), S.d_gpts.as<float4>(), S.d_gmask.as<uint8_t>(), recs, eig_list, flags + 3);
                check_launch("refine_h_kernel<0>");
                // the small candidates' LM in the eigen kernel's lanes: less wave time (3.7 -> 2.1 s per headline unit) but a longer
                // critical path (one lane's ten iterations, ~2 ms) — it pays when the candidates outnumber the resident waves
                const char* lane_env = getenv("SLIDEO_REFINE_LANE_LM");                 // (read per unit: the tests switch it)
                const int lane_lm = lane_env ? atoi(lane_env) : (ncand >= 4096 ? 1 : 0);
                refine_h_eigen_kernel<<<cdiv((int)ncand, HJ), 64, refine_h_eigen_lds_bytes(), st>>>(
                    vp, S.d_qofs.as<uint32_t>(), S.d_kp.as<slideo_keypoint>(), m->d_page_xy.as<float2>(), S.d_votes.as<uint2>(),
                    S.d_fcs.as<FrameCands>(), S.d_gmask.as<uint8_t>(), c.max_candidate_pages, recs, eig_list, flags + 3, lane_lm);
                check_launch("refine_h_eigen_kernel");
                refine_h_kernel<1><<<dim3(c.max_candidate_pages, n), 64, 0, st>>>(
                    vp, S.d_qofs.as<uint32_t>(), S.d_kp.as<slideo_keypoint>(), m->d_page_xy.as<float2>(), S.d_votes.as<uint2>(),
                    S.d_fcs.as<FrameCands>(), S.d_gpts.as<float4>(), S.d_gmask.as<uint8_t>(), recs, eig_list, flags + 3);
                check_launch("refine_h_kernel<1>");
            }
        } else {
        ransac_kernel<RANSAC_SMALL_PTS, 0><<<dim3(c.max_candidate_pages, n), 64, 0, st>>>(
            vp, S.d_qofs.as<uint32_t>(), S.d_kp.as<slideo_keypoint>(), m->d_page_xy.as<float2>(), S.d_votes.as<uint2>(),
            m->d_rng.as<uint32_t>(), S.d_fcs.as<FrameCands>(), S.d_gpts.as<float4>(), S.d_gmask.as<uint8_t>(), flags);
        check_launch("ransac_kernel (small)");
        ransac_kernel<RANSAC_LDS_PTS, RANSAC_SMALL_PTS + 1><<<dim3(c.max_candidate_pages, n), 64, 0, st>>>(
            vp, S.d_qofs.as<uint32_t>(), S.d_kp.as<slideo_keypoint>(), m->d_page_xy.as<float2>(), S.d_votes.as<uint2>(),
            m->d_rng.as<uint32_t>(), S.d_fcs.as<FrameCands>(), S.d_gpts.as<float4>(), S.d_gmask.as<uint8_t>(), flags);
        check_launch("ransac_kernel (large)");
        }
        uint32_t* pair_count = S.d_pairs.as<uint32_t>();
        PairDesc* pair_list = reinterpret_cast<PairDesc*>(S.d_pairs.as<uint8_t>() + 64);
        HIP_CHECK(hipMemsetAsync(pair_count, 0, 16, st));
        rate_kernel<<<n, 64, 0, st>>>(vp, n, S.d_fcs.as<FrameCands>(), m->d_pageinfo.as<PageInfo>(), pair_list, pair_count);
        check_launch("rate_kernel");
        int max_tile_rows = 0;
        for (const AreaGeom& ag : m->area_geoms) max_tile_rows = std::max(max_tile_rows, cdiv(ag.dh, SM_TH));
        if (c.verify_model == 1)
            reproject_kernel<true><<<dim3(max_tile_rows, std::min(n, 65535)), 256, 0, st>>>(m->d_area_geoms.as<AreaGeom>(), m->d_area_taps.as<AreaTap>(),
                                                                                            m->d_area_idx.as<int32_t>(),
                                                                                            m->d_page_small.as<uint8_t>(), frames_dev, frame_stride,
                                                                                            stride, w, h, S.d_fcs.as<FrameCands>(), pair_list, pair_count);
        else
            reproject_kernel<false><<<dim3(max_tile_rows, std::min(n, 65535)), 256, 0, st>>>(m->d_area_geoms.as<AreaGeom>(), m->d_area_taps.as<AreaTap>(),
                                                                                             m->d_area_idx.as<int32_t>(),
                                                                                             m->d_page_small.as<uint8_t>(), frames_dev, frame_stride,
                                                                                             stride, w, h, S.d_fcs.as<FrameCands>(), pair_list, pair_count);
        check_launch("reproject_kernel");
    }
    verdict_kernel<<<cdiv(n, 64), 64, 0, st>>>(vp, n, S.d_qofs.as<uint32_t>(), m->d_pageinfo.as<PageInfo>(), S.d_fcs.as<FrameCands>(),
                                               S.d_verdicts.as<slideo_verdict>());
    check_launch("verdict_kernel");
    if (prof) HIP_CHECK(hipEventRecord(S.ev[3], st));
    uint8_t* ho = S.h_out.as<uint8_t>();
    const size_t tail = (size_t)n * (sizeof(slideo_verdict) + sizeof(FrameCands));
    HIP_CHECK(hipMemcpyAsync(ho, S.d_verdicts.p, (size_t)n * sizeof(slideo_verdict), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(ho + (size_t)n * sizeof(slideo_verdict), S.d_fcs.p, (size_t)n * sizeof(FrameCands), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(ho + tail, flags, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(ho + tail + 4, S.d_info.p, 8, hipMemcpyDeviceToHost, st));      // {Qtot, max keypoints per frame}
    if (prof) HIP_CHECK(hipEventRecord(S.ev[4], st));
    S.busy = true; S.n = n;
}
void unit_submit_sift(slideo_matcher* m, Slot& S, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride);
void add_pages_sift(slideo_matcher* m, Slot& S, int cnt, int w, int h, int stride, int64_t fb);

void unit_submit(slideo_matcher* m, Slot& S, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride,
                 bool allow_async = true) {
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
    const bool async = allow_async && m->async_submit && knn_engine_for(m, n * (int)std::min<uint32_t>(kpcap, (uint32_t)c.nfeatures)) != 1 &&
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
    // (the matrix-core engine searches the unique rows of the train set, the VALU engine — A/B only — all of them)
    const bool dedup = m->Mu < m->M && knn_engine_for(m, (int)qplan) != 1;
    const int nt_knn = (int)(dedup ? m->Mu : m->M);
    S.u_nt = nt_knn;
    knn_reserve(m, S, (int)qplan, nt_knn, (int)qtot);
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
    if (m->knn_chain && m->last_knn_ev && m->last_knn_ev != S.ev_knn) HIP_CHECK(hipStreamWaitEvent(st, m->last_knn_ev, 0));
    if (prof) HIP_CHECK(hipEventRecord(S.ev[1], st));
    if (qtot > 0) {
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
        run_knn(m, S, S.d_desc.as<uint32_t>(), (int)qplan, T, nt_knn, prune, async ? S.d_qofs.as<uint32_t>() + n : nullptr, (int)qtot);
        if (prof) HIP_CHECK(hipEventRecord(S.ev[2], st));      // the kNN interval ends here: the search kernel (+ its segment merge)
        if (m->knn_chain) { HIP_CHECK(hipEventRecord(S.ev_knn, st)); m->last_knn_ev = S.ev_knn; }
        if (dedup) {
            knn_expand_dups_kernel<KLIST><<<cdiv((int)std::max(qtot, 1u), KNN_BLOCK), KNN_BLOCK, 0, st>>>(
                S.d_keys.as<uint32_t>(), (int)qtot, m->d_grp_next.as<int32_t>(), async ? S.d_qofs.as<uint32_t>() + n : nullptr);
            check_launch("knn_expand_dups_kernel");
        }
    }
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
    } catch (...) {
        for (Slot& S : m->slots) { (void)hipStreamSynchronize(S.st); S.busy = false; }
        throw;
    }
}

void set_err(slideo_matcher* m, const char* what) {
    if (m) m->err = what;
    else { std::lock_guard<std::mutex> lk(g_err_mutex); g_create_error = what; }
}

}  // namespace

#define API_TRY try {
#define API_CATCH(m)                                                          \
    }                                                                         \
    catch (const slideo::Error& e) { set_err(m, e.what()); return e.code; }   \
    catch (const std::exception& e) { set_err(m, e.what()); return SLIDEO_ERR_HIP; } \
    catch (...) { set_err(m, "unknown error"); return SLIDEO_ERR_HIP; }       \
    return SLIDEO_OK;

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
    std::memset(&c->ocv, 0, sizeof(c->ocv));         // every OpenCV-variant switch at its default
    c->ocv.rng_mul = 4164903690u;                    // CV_RNG_COEFF
}

const char* slideo_last_error(const slideo_matcher* m) {
    if (m) return m->err.c_str();
    std::lock_guard<std::mutex> lk(g_err_mutex);
    static thread_local std::string copy;
    copy = g_create_error;
    return copy.c_str();
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
    if (const char* e = std::getenv("SLIDEO_KNN_ENGINE")) { const int v = std::atoi(e); if (v >= 0 && v <= 3) mm->knn_engine = v; }
    if (const char* e = std::getenv("SLIDEO_ASYNC_SUBMIT")) mm->async_submit = std::atoi(e) != 0;
    if (const char* e = std::getenv("SLIDEO_KNN_DEDUP")) mm->knn_dedup = std::atoi(e) != 0;
    if (const char* e = std::getenv("SLIDEO_LSH_ENGINE")) mm->lsh_gather = std::string(e) == "gather";
    if (const char* e = std::getenv("SLIDEO_HOST_UNIT")) mm->host_unit = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("SLIDEO_ORB_CHAIN")) mm->orb_chain = std::atoi(e) != 0;
    if (const char* e = std::getenv("SLIDEO_KNN_CHAIN")) mm->knn_chain = std::atoi(e) != 0;
    if (const char* e = std::getenv("SLIDEO_PYR_CHAIN")) mm->pyr_chain = std::atoi(e) != 0;
    if (const char* e = std::getenv("SLIDEO_WS_GB")) { double gb = std::atof(e); if (gb > 0.1) mm->ws_budget = (size_t)(gb * (double)((size_t)1 << 30)); }
    for (Slot& S : mm->slots) {
        HIP_CHECK(hipStreamCreateWithFlags(&S.st, hipStreamNonBlocking));
        for (auto& e : S.ev) HIP_CHECK(hipEventCreate(&e));
        HIP_CHECK(hipEventCreateWithFlags(&S.ev_in, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&S.ev_orb, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&S.ev_knn, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&S.ev_up, hipEventDisableTiming));
    }
    HIP_CHECK(hipStreamCreateWithFlags(&mm->copy_st, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&mm->sift_ev, hipEventDisableTiming));
    mm->stream = mm->slots[0].st;
    OrbTables t{};
    umax_table(cfg->patch_size / 2, t.umax);
    if (cfg->ocv.blur == 2) gauss7_q8_rounded(t.gk); else gauss7_fixed(t.gk);
    gauss7_f32(t.gkf);
    brief_pattern(cfg->patch_size, t.pattern, cfg->ocv.rng_mul);
    mm->d_tables.reserve(sizeof(OrbTables));
    HIP_CHECK(hipMemcpy(mm->d_tables.p, &t, sizeof(t), hipMemcpyHostToDevice));
    {
        std::vector<uint32_t> ict;
        ic_weight_table(cfg->patch_size / 2, t.umax, ict, mm->ic_shift);
        mm->ic_entries = (int)(ict.size() / 2);
        mm->d_ictab.reserve(ict.size() * 4);
        HIP_CHECK(hipMemcpy(mm->d_ictab.p, ict.data(), ict.size() * 4, hipMemcpyHostToDevice));
    }
    {
        // similarity: 2 draws per iteration + redraws; homography: 4 per attempt, several attempts per accepted subset
        int64_t len = std::max<int64_t>(RNG_TABLE_MIN, (cfg->verify_model == 1 ? 64ll : 4ll) * std::max(cfg->ransac_max_iters, 1) + 2048);
        if (const char* e = getenv("SLIDEO_RNG_STREAM_LEN")) len = std::max<int64_t>(512, atoll(e));     // tests: force the growth path
        upload_rng_stream(mm.get(), (uint32_t)std::min<int64_t>(len, 1ll << 26));
    }
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&vote_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&describe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  describe_window(cfg->patch_size / 2).dwords * 16));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ransac_h_kernel<RANSAC_SMALL_PTS, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)ransac_h_lds_bytes(RANSAC_SMALL_PTS, 0)));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ransac_h_kernel<RANSAC_LDS_PTS, RANSAC_SMALL_PTS + 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)ransac_h_lds_bytes(RANSAC_LDS_PTS, 0)));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ransac_h_kernel<RANSAC_SMALL_PTS, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)ransac_h_lds_bytes(RANSAC_SMALL_PTS, 1)));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ransac_h_kernel<RANSAC_LDS_PTS, RANSAC_SMALL_PTS + 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)ransac_h_lds_bytes(RANSAC_LDS_PTS, 1)));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ransac_h_kernel<RANSAC_SMALL_PTS, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)ransac_h_lds_bytes(RANSAC_SMALL_PTS, 2)));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ransac_h_kernel<RANSAC_LDS_PTS, RANSAC_SMALL_PTS + 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)ransac_h_lds_bytes(RANSAC_LDS_PTS, 2)));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&refine_h_eigen_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)refine_h_eigen_lds_bytes()));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ransac_h_tail_kernel<1, RANSAC_H_TAIL_WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)ransac_h_tail_lds_bytes(RANSAC_H_TAIL_WAVES)));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ransac_h_tail_kernel<2, RANSAC_H_TAIL_WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)ransac_h_tail_lds_bytes(RANSAC_H_TAIL_WAVES)));
    m = mm.release();
    *out = m;
    API_CATCH(nullptr)
}

void slideo_matcher_destroy(slideo_matcher* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    for (Slot& S : m->slots) {
        if (S.st) { (void)hipStreamSynchronize(S.st); (void)hipStreamDestroy(S.st); }
        for (auto& e : S.ev) if (e) (void)hipEventDestroy(e);
        if (S.ev_in) (void)hipEventDestroy(S.ev_in);
        if (S.ev_orb) (void)hipEventDestroy(S.ev_orb);
        if (S.ev_knn) (void)hipEventDestroy(S.ev_knn);
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
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    hipStream_t st = S.st;
    const uint64_t total = (uint64_t)n_pages;
    if (m->progress) m->progress(m->progress_user, 0, total, "Analyzing PDF pages...");     // lib.rs:43
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
        S.d_stage.reserve(fb * cnt + 16);
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
            m->pages.push_back(std::move(pg));
            if (m->progress) m->progress(m->progress_user, (uint64_t)(i + j + 1), total, "Analyzing PDF pages...");   // lib.rs:49-53
        }
        i += cnt;
    }
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

void l2_prepare(slideo_matcher::L2Set& L, const uint8_t* t, int nt, hipStream_t st);      // (defined with the L2 entry points below)

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
    m->d_train.reserve(std::max<size_t>(train.size(), 64) + 64);   // + slack: the kNN loop reads whole rows only, no overrun
    m->d_train_page.reserve(std::max<size_t>(tpage.size() * 4, 16));
    m->d_page_xy.reserve(std::max<size_t>(xy.size() * sizeof(float2), 16));
    m->d_pageinfo.reserve(std::max<size_t>(info.size() * sizeof(PageInfo), 16));
    m->d_page_small.reserve(std::max<size_t>(smalls.size(), 16));
    if (M > 0 && m->sift_on) {
        // SIFT mode: the rows become the train set of the squared-L2 engine (norm order, centred tile-major operand)
        HIP_CHECK(hipMemcpy(m->d_train_page.p, tpage.data(), tpage.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(m->d_page_xy.p, xy.data(), xy.size() * sizeof(float2), hipMemcpyHostToDevice));
        l2_prepare(m->l2, train.data(), (int)M, m->stream);
        m->Mu = M;
    } else if (M > 0) {
        HIP_CHECK(hipMemcpy(m->d_train.p, train.data(), train.size(), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(m->d_train_page.p, tpage.data(), tpage.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(m->d_page_xy.p, xy.data(), xy.size() * sizeof(float2), hipMemcpyHostToDevice));
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
    m->kept.valid = false;
    upload_frames(S, frames, n_frames, height, stride_bytes, frame_stride_bytes);
    m->kept = slideo_matcher::Kept{true, n_frames, width, height, stride_bytes};   // stays in slot 0's staging buffer: slideo_match_kept_frames
    run_small(m, S.d_stage.as<uint8_t>(), n_frames, width, height, stride_bytes, (int64_t)fb, sw, sh, st);
    const size_t sb = (size_t)sw * sh * 3;
    DevBuf& prev = m->d_prev_small;
    prev.reserve(sb);
    if (prev_small) HIP_CHECK(hipMemcpyAsync(prev.p, prev_small, sb, hipMemcpyHostToDevice, st));
    m->d_ssd.reserve((size_t)n_frames * 8);
    // pair i: (small[i-1], small[i]); pair 0 uses prev
    if (prev_small) {
        ssd_kernel<<<1, 256, 0, st>>>(prev.as<uint8_t>(), 0, m->d_small.as<uint8_t>(), 0, (int64_t)sb, m->d_ssd.as<unsigned long long>());
        check_launch("ssd_kernel");
    }
    if (n_frames > 1) {
        ssd_kernel<<<n_frames - 1, 256, 0, st>>>(m->d_small.as<uint8_t>(), (int64_t)sb, m->d_small.as<uint8_t>() + sb, (int64_t)sb, (int64_t)sb,
                                                 m->d_ssd.as<unsigned long long>() + 1);
        check_launch("ssd_kernel");
    }
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

// ---- debug taps ---------------------------------------------------------------------------

int32_t slideo_orb_bgr8(slideo_matcher* m, const uint8_t* bgr, int32_t width, int32_t height, int32_t stride_bytes,
                        slideo_keypoint* kp, uint8_t* desc32, int32_t capacity, int32_t* n_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (!bgr || !n_out) fail(SLIDEO_ERR_INVALID_ARG, "null image/n_out");
    validate_image(width, height, stride_bytes);
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    hipStream_t st = S.st;
    const size_t fb = (size_t)height * stride_bytes;
    S.d_stage.reserve(fb + 16);
    HIP_CHECK(hipMemcpyAsync(S.d_stage.p, bgr, fb, hipMemcpyHostToDevice, st));
    run_orb(m, S, S.d_stage.as<uint8_t>(), 1, width, height, stride_bytes, (int64_t)fb, false);
    const uint32_t q = S.orb.qtot;
    *n_out = (int32_t)q;
    if ((int64_t)q > capacity) fail(SLIDEO_ERR_CAPACITY, "%u keypoints, capacity %d", q, capacity);
    if (q) {
        if (kp) HIP_CHECK(hipMemcpyAsync(kp, S.d_kp.p, (size_t)q * sizeof(slideo_keypoint), hipMemcpyDeviceToHost, st));
        if (desc32) HIP_CHECK(hipMemcpyAsync(desc32, S.d_desc.p, (size_t)q * 32, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    API_CATCH(m)
}

int32_t slideo_pyramid_level_bgr8(slideo_matcher* m, const uint8_t* bgr, int32_t width, int32_t height, int32_t stride_bytes,
                                  int32_t level, int32_t blurred, uint8_t* out, int64_t out_capacity, int32_t* lw, int32_t* lh) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (!bgr || !out || !lw || !lh) fail(SLIDEO_ERR_INVALID_ARG, "null argument");
    validate_image(width, height, stride_bytes);
    if (level < 0 || level >= m->cfg.nlevels) fail(SLIDEO_ERR_INVALID_ARG, "level out of range");
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    hipStream_t st = S.st;
    const size_t fb = (size_t)height * stride_bytes;
    S.d_stage.reserve(fb + 16);
    HIP_CHECK(hipMemcpyAsync(S.d_stage.p, bgr, fb, hipMemcpyHostToDevice, st));
    run_orb(m, S, S.d_stage.as<uint8_t>(), 1, width, height, stride_bytes, (int64_t)fb, false, blurred != 0);
    const LevelGeom& L = geom_for(m, width, height).g.lv[level];
    *lw = L.w; *lh = L.h;
    if ((int64_t)L.w * L.h > out_capacity) fail(SLIDEO_ERR_CAPACITY, "level needs %lld bytes", (long long)L.w * L.h);
    if (L.w > 0 && L.h > 0) {
        const uint8_t* src = (blurred ? S.d_blur.as<uint8_t>() : S.d_pyr.as<uint8_t>()) + L.ofs;
        HIP_CHECK(hipMemcpy2DAsync(out, L.w, src, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    API_CATCH(m)
}

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
              DevBuf* keys = nullptr, DevBuf* pend = nullptr, float prune_tol = 0.f) {
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

// ---- SIFT (csrc/sift.hip.h) --------------------------------------------------------------------------------------------
extern "C++" {
namespace {

SiftGeom sift_geom(int w, int h) {
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

SiftTaps sift_taps(double sigma) {          // getGaussianKernel(cvRound(8 sigma + 1) | 1, sigma) in f32 (oracle sift_gauss_kernel)
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
bool sift_blur_fast(bool fma, const float* src, int64_t sf, float* dst, int64_t df, float* dog, int64_t dgf, int w, int h, int n, const SiftTaps& tp, hipStream_t st,
                    float* half, int64_t hf, int hw, int hh) {      // returns: the half-size copy was written too
    static const bool tiles = std::getenv("SLIDEO_SIFT_BLUR_TILES") != nullptr;                   // A/B: the 64 x 64 tile kernel
    if (tiles) {
        const dim3 grid(cdiv(w, 64) * cdiv(h, 64), n);
        if (fma) sift_blur_fast_kernel<N, true><<<grid, 256, 0, st>>>(src, sf, dst, df, dog, dgf, w, h, tp);
        else sift_blur_fast_kernel<N, false><<<grid, 256, 0, st>>>(src, sf, dst, df, dog, dgf, w, h, tp);
        return false;
    }
    // streams: one wave per (64-column strip, chunk of rows); chunks sized so that a launch has ~16 k waves (several rounds of the chip: a short tail)
    const int strips = cdiv(w, 64);
    static const int target_waves = [] { const char* e = std::getenv("SLIDEO_SIFT_WAVES"); return e ? std::max(atoi(e), 1) : 16384; }();
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
bool sift_blur_launch(slideo_matcher* m, const float* src, int64_t sf, float* dst, int64_t df, float* dog, int64_t dgf, int w, int h, int n,
                      const SiftTaps& tp, hipStream_t st, float* half = nullptr, int64_t hf = 0, int hw = 0, int hh = 0) {
    bool half_done = false;
    const bool fma = m->cfg.ocv.blur != 1;
    static const bool generic_only = std::getenv("SLIDEO_SIFT_BLUR_GENERIC") != nullptr;        // A/B and the equality test
    bool fast = !generic_only && w >= 32 && h >= 32;
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
void sift_pyramids(slideo_matcher* m, const uint8_t* frames_dev, int nb, int w, int h, int stride, int64_t fs, const slideo_sift_config& sc,
                   const SiftGeom& g, hipStream_t st) {
    auto& W = m->sift;
    const int64_t base_frame = (int64_t)g.ow[0] * g.oh[0];
    W.base.reserve((size_t)base_frame * nb * 4);
    W.gauss.reserve((size_t)g.g_frame * nb * 4 + 64);
    const GrayCoef gc = m->cfg.ocv.gray == 1 ? GrayCoef{1868u, 9617u, 4899u, 14u} : GrayCoef{3735u, 19235u, 9798u, 15u};
    {
        // gray u8 first (the ORB path's kernel), then the doubled f32 base image from it (the DoG
        // pyramid itself is not stored: sift.hip.h SiftDog)
        const int gp = ((w + 15) & ~15) + 16;
        const int64_t gframe = (int64_t)gp * h;
        W.gray.reserve((size_t)gframe * nb + 64);
        const int aligned4 = ((uintptr_t)frames_dev % 4 == 0) && (stride % 4 == 0) && (fs % 4 == 0);
        gray_kernel<<<dim3(cdiv(cdiv(w, 4), 256), h, nb), 256, 0, st>>>(frames_dev, fs, stride, W.gray.as<uint8_t>(), gframe, w, h, gp, aligned4, gc);
        check_launch("gray_kernel");
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
    static const bool fuse_half = std::getenv("SLIDEO_SIFT_FUSE_HALF") == nullptr || atoi(std::getenv("SLIDEO_SIFT_FUSE_HALF")) != 0;   // A/B
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
            const bool want_half = fuse_half && i == SIFT_NL && o + 1 < g.n_oct;
            const bool did = sift_blur_launch(m, G + g.g_ofs[o] + lsz * (i - 1), g.g_frame, G + g.g_ofs[o] + lsz * i, g.g_frame, nullptr, 0,      // (the DoG layers are not stored: SiftDog)
                                              ow, oh, nb, sift_taps(sig[i]), st,
                                              want_half ? G + g.g_ofs[o + 1] : nullptr, g.g_frame, want_half ? g.ow[o + 1] : 0, want_half ? g.oh[o + 1] : 0);
            if (want_half) half_done = did;
        }
    }
}

// SIFT of n device frames: results appended at kp_dev / desc_dev from row `row0`; per-frame counts into counts_host[0..n).
// Returns the rows written.
int64_t sift_batch(slideo_matcher* m, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t fs, const slideo_sift_config& sc,
                   int64_t row0, int64_t capacity_total, slideo_keypoint* kp_dev, uint8_t* desc_dev, uint32_t* counts_host, hipStream_t st) {
    auto& W = m->sift;
    const SiftGeom g = sift_geom(w, h);
    SiftParams sp{};
    sp.nfeatures = sc.nfeatures; sp.cand_cap = 1 << 16; sp.raw_cap = 1 << 15;
    sp.contrast_threshold = (float)sc.contrast_threshold; sp.edge_threshold = (float)sc.edge_threshold; sp.sigma = (float)sc.sigma;
    sp.threshold = (int)std::floor(0.5 * sc.contrast_threshold / SIFT_NL * 255);
    sp.atan_fma = m->cfg.ocv.atan; sp.blur_fma = m->cfg.ocv.blur != 1;
    // frames per pass under a 24 GB budget for the pyramids (265 MB + 33 MB per 1080p frame; SLIDEO_SIFT_WS_MB: tests)
    const size_t per = ((size_t)g.g_frame + (size_t)g.ow[0] * g.oh[0]) * 4;
    const char* budget_env = std::getenv("SLIDEO_SIFT_WS_MB");          // (read per call: a test squeezes it)
    const size_t budget = budget_env ? (size_t)std::max(atoll(budget_env), 1ll) << 20 : (size_t)24 << 30;
    const int nb_max = (int)std::max<size_t>(1, budget / std::max<size_t>(per, 1));
    int64_t rows = 0;
    for (int f0 = 0; f0 < n; f0 += nb_max) {
        const int nb = std::min(nb_max, n - f0);
        sift_pyramids(m, frames_dev + (int64_t)f0 * fs, nb, w, h, stride, fs, sc, g, st);
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
        std::vector<uint32_t> hc((size_t)nb * 3);
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
        scan_kernel<<<1, 1024, 0, st>>>(kept_count, nb, W.qofs.as<uint32_t>(), W.info.as<uint32_t>());
        check_launch("scan_kernel");
        uint32_t info[6];
        HIP_CHECK(hipMemcpyAsync(info, W.info.p, 24, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(hc.data(), kept_count, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (info[4] & 16u) fail(SLIDEO_ERR_CAPACITY, "SIFT: more than %d scale-space extrema in a frame", sp.cand_cap);
        if (info[4] & 32u) fail(SLIDEO_ERR_CAPACITY, "SIFT: more than %d keypoints in a frame before retainBest", sp.raw_cap);
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
int64_t sift_unit_capacity(const slideo_matcher* m, int n) {
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
        static const bool l2_prune = std::getenv("SLIDEO_L2_PRUNE") == nullptr || std::atoi(std::getenv("SLIDEO_L2_PRUNE")) != 0;
        l2_query(m, m->l2, S.d_desc.as<uint8_t>(), (int)qtot, kq, st, S, false, &S.d_blur, &S.d_knn_pend, (!lowe && l2_prune) ? c.vote_tolerance : 0.f);
        const int kl = kq <= 8 ? 8 : (kq <= 16 ? 16 : KLIST);                 // (the list length of the instance l2_query picked)
        if (lowe)
            l2_ratio_keys_kernel<<<cdiv((int)qtot, 256), 256, 0, st>>>(S.d_blur.as<unsigned long long>(), kl, (int)qtot, m->sift_ratio,
                                                                      S.d_keys.as<uint32_t>(), KLIST);
        else
            l2_tol_keys_kernel<<<cdiv((int)qtot, 256), 256, 0, st>>>(S.d_blur.as<unsigned long long>(), kl, kq, (int)qtot, c.vote_tolerance,
                                                                    S.d_keys.as<uint32_t>(), KLIST);
        check_launch("l2 keys kernel");
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

}  // namespace
}  // extern "C++"

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
    m->kept.valid = false;
    S.d_stage.reserve(fb + 16);
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
    m->kept.valid = false;
    S.d_stage.reserve(fb + 16);
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

int32_t slideo_small_image_bgr8(slideo_matcher* m, const uint8_t* bgr, int32_t width, int32_t height, int32_t stride_bytes,
                                uint8_t* out, int64_t out_capacity, int32_t* sw_out, int32_t* sh_out) {
    if (!m) return SLIDEO_ERR_INVALID_ARG;
    API_TRY
    if (!bgr || !out || !sw_out || !sh_out) fail(SLIDEO_ERR_INVALID_ARG, "null argument");
    validate_image(width, height, stride_bytes);
    HIP_CHECK(hipSetDevice(m->device));
    require_idle(m);
    Slot& S = m->slots[0];
    hipStream_t st = S.st;
    const size_t fb = (size_t)height * stride_bytes;
    S.d_stage.reserve(fb + 16);
    HIP_CHECK(hipMemcpyAsync(S.d_stage.p, bgr, fb, hipMemcpyHostToDevice, st));
    int sw = 0, sh = 0;
    run_small(m, S.d_stage.as<uint8_t>(), 1, width, height, stride_bytes, (int64_t)fb, sw, sh, st);
    *sw_out = sw; *sh_out = sh;
    if ((int64_t)sw * sh * 3 > out_capacity) fail(SLIDEO_ERR_CAPACITY, "small image needs %lld bytes", (long long)sw * sh * 3);
    HIP_CHECK(hipMemcpyAsync(out, m->d_small.p, (size_t)sw * sh * 3, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    API_CATCH(m)
}

}  // extern "C"
