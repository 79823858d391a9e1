// runtime.hpp — the host runtime behind include/slideo_amd.h, shared by its translation units.
//
//   capi_runtime.hip   handles, page database, slots, unit submit / collect, the match entry points
//   stage_orb.hip      ORB stage drivers          (kernels: orb.hip.h)
//   stage_knn.hip      index build + k-NN stage   (kernels: knn.hip.h, knn_tile.hip.h, knn_l2.hip.h, knn_lsh.hip.h)
//   stage_verify.hip   vote .. verdict, small img (kernels: verify.hip.h, homography.hip.h)
//   stage_sift.hip     SIFT stage + entry points  (kernels: sift.hip.h)
//   capi_taps.hip      debug taps of the parity tests
//   capi_group.cpp     the N-device group (slideo_group_*)
//
// Every kernel header is compiled by exactly one unit; what crosses units is the functions declared below and the plain
// records of types.h.  There is no CPU fallback anywhere in here.
#pragma once
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
#include "slideo_amd.h"
#include "types.h"

namespace slideo {

#ifndef SLIDEO_NSLOTS
#define SLIDEO_NSLOTS 4
#endif
constexpr int NSLOTS = SLIDEO_NSLOTS;      // units in flight (each with its own workspace and HIP stream)
constexpr int KLIST = 32;
static_assert(KLIST == VOTE_KLIST, "vote_kernel reads whole key lists");

// Environment switches (all listed in include/slideo_amd.h, "Environment").  None changes a result.
inline long env_long(const char* name, long dflt) {
    const char* e = std::getenv(name);
    return e && *e ? std::atol(e) : dflt;
}

struct GeomEntry {
    int w = 0, h = 0;
    PyrGeom g;
    DevBuf lin_tab;
    DevBuf fast_tiles;                // per FAST tile: level, origin, raw-column alignment (orb.hip.h fast_tile_entry)
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

// One workspace + stream.  Several slots let the ORB stage of one unit of frames run concurrently with the
// kNN / verification stages of the previous units (matrix-core bound vs VALU/LDS/HBM bound work).
struct Slot {
    hipStream_t st = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_in = nullptr, ev_orb = nullptr, ev_up = nullptr;
    // SLIDEO_CU_SPLIT (measurement switch, off by default): the search on a stream of its own whose CU mask holds N CUs, the other
    // stages on the complement — spatial instead of per-CU sharing; ev_k0 / ev_k1 order the search stream behind / before the slot's
    hipStream_t st_knn = nullptr;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;
    // arguments of the unit in flight (re-run through the exact-size path if the capacity-sized one overflowed)
    const uint8_t* u_frames = nullptr; int u_w = 0, u_h = 0, u_stride = 0; int64_t u_fs = 0; bool u_async = false; int u_nt = 0; bool u_shared = false, u_w12 = false;
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
        // (d_stage is NOT in the list: slot 0's holds the frames slideo_changed_mask_bgr8 kept, and host-frame units size it
        // themselves before their copy)
        DevBuf* mine[] = {&d_pyr, &d_blur, &d_cand, &d_hist, &d_candcount, &d_flags, &d_thr, &d_lvlofs, &d_kpcount, &d_qofs,
                          &d_info, &d_items, &d_kp, &d_desc, &d_keys, &d_knn_pend, &d_votes, &d_gpts, &d_gmask, &d_fcs, &d_verdicts, &d_pairs, &d_blurmask,
                          &d_qkeys, &d_tail, &d_refine};
        const DevBuf* theirs[] = {&o.d_pyr, &o.d_blur, &o.d_cand, &o.d_hist, &o.d_candcount, &o.d_flags, &o.d_thr, &o.d_lvlofs,
                                  &o.d_kpcount, &o.d_qofs, &o.d_info, &o.d_items, &o.d_kp, &o.d_desc, &o.d_keys, &o.d_knn_pend, &o.d_votes,
                                  &o.d_gpts, &o.d_gmask, &o.d_fcs, &o.d_verdicts, &o.d_pairs, &o.d_blurmask, &o.d_qkeys, &o.d_tail, &o.d_refine};
        static_assert(sizeof(mine) / sizeof(mine[0]) == sizeof(theirs) / sizeof(theirs[0]), "same buffer lists");
        for (size_t i = 0; i < sizeof(mine) / sizeof(mine[0]); ++i) mine[i]->reserve_cap(theirs[i]->cap);
        h_info.reserve_cap(o.h_info.cap); h_out.reserve_cap(o.h_out.cap);
    }
};

}  // namespace slideo

struct slideo_matcher {
    slideo_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;      // = slots[0].st (setup, page ingest, taps)
    std::string err;
    slideo_progress_fn progress = nullptr;
    void* progress_user = nullptr;
    size_t ws_budget = (size_t)48 << 30;      // all slots together (SLIDEO_WS_GB); 288 GB of HBM per GPU
    long sift_ws_mb = 24l << 10;              // pyramids of one SIFT pass (SLIDEO_SIFT_WS_MB); 96 GB on a device with >= 192 GB: 256 1080p frames in ONE pass

    slideo::DevBuf d_tables, d_rng, d_ictab;
    struct L2Set { slideo::DevBuf d_tx, d_tn, d_side, d_perm, d_keys, d_pend; int nt = 0, nt_pad = 0; bool ready = false; } l2;   // cfg2: the L2 train set
    uint32_t rng_len = 0;
    int ic_shift = 0, ic_entries = 0;     // intensity-centroid weight table of describe_kernel (geom.h ic_weight_table)
    std::vector<std::unique_ptr<slideo::GeomEntry>> geoms;

    // INTER_AREA size classes
    std::vector<slideo::AreaGeom> area_geoms;
    std::vector<slideo::AreaTap> area_taps;
    std::vector<int32_t> area_idx;
    std::vector<slideo::AreaRec> area_recs;
    slideo::DevBuf d_area_geoms, d_area_taps, d_area_idx, d_area_recs;
    bool area_dirty = true;

    // pages
    std::vector<slideo::HostPage> pages;
    bool finalized = false;
    // slideo_matcher_use_sift: SIFT features + L2 k-NN + ratio test / tolerance vote in front of the verify stage.  The SIFT and L2
    // workspaces are the matcher's (not a slot's): the extraction stages of consecutive units take turns (sift_ev)
    bool sift_on = false;
    slideo_sift_config sift_cfg{};
    float sift_ratio = 0.f;
    hipEvent_t sift_ev = nullptr;
    bool sift_ev_set = false;
    int64_t M = -1;
    slideo::DevBuf d_train, d_train_page, d_page_xy, d_pageinfo, d_page_small;
    slideo::DevBuf d_trainb, d_train_side, d_train_nminh, d_train_perm;   // {0,1} FP4 operand in norm order + its side arrays (knn_tile.hip.h)
    // train-set de-duplication (knn.hip.h knn_expand_dups_kernel): the matrix-core engine searches the Mu unique rows, keys carry
    // the lowest original row of a group, d_grp_next chains the equal rows.  SLIDEO_KNN_DEDUP=0 searches all M rows.
    slideo::DevBuf d_utrain, d_grp_next;
    struct LshSet { slideo::DevBuf ofs, rows, keys; slideo::LshDev dev{}; bool ready = false; } lsh;      // slideo_config.matcher 1 (knn_lsh.hip.h)
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
    slideo::DevBuf d_kept;
    bool units_pending = false;   // the call being served has more units than the one submitted now
    int cu_split = 0;       // SLIDEO_CU_SPLIT=N: the search on N CUs (two blocks per CU), ORB / verify on the other 256 - N (0 = off: every stream on every CU)
    // while units share the chip the search runs the 12-wave block (three waves per SIMD, 128 registers each) instead of the 8-wave
    // block + LDS pad when a unit carries at least this many (query, train row) pairs per frame pixel: the larger the deck, the more of
    // a step is search, and from ~290 pairs per pixel on the fuller matrix pipe is worth more than the co-runners' occupancy
    // (profiles/r06_experiments.txt 7: headline 197: - 2..4 %; 700 pages 275: - 1.7 %; 800 pages 314: + 5.5 %; configs[3] 392:
    // + 6.8 %; configs[4] 352: + 4.7 %).  SLIDEO_KNN_W12_RATIO overrides (0 = never).
    double knn_w12_ratio = 290.0;
    int knn_nseg_force = 0; // SLIDEO_KNN_NSEG=n (measurement): train-stream segments of the matrix-core search instead of the plan's
    int knn_share = -1;     // search blocks per CU: -1 = one while other units are in flight, two otherwise (default); 0 = always two; 1 = always one; 3 / 4 = the 12-wave block while shared / always; 5 / 6 = the 1-tile 12-wave block (knn_tile1.hip.h) while shared / always (SLIDEO_KNN_SHARE)
    int knn_engine = 0;     // 0 = FP4 MFMA, wave shape chosen per launch (default), 1 = integer VALU popcount,
                            // 2 = FP4 MFMA, 2 waves/SIMD x 4 query tiles (knn_tile4_kernel), 3 = 4 waves/SIMD x 2 tiles (knn_tile2_kernel)
    int knn_exact_lists = 0;  // 1 = the matcher's kNN stage keeps full exact k-NN lists (no fused vote filter)
    // Units are enqueued in one go, without the mid-unit host wait for the keypoint counts: everything downstream of the ORB
    // counts is sized by capacity and reads the counts on the device (SLIDEO_ASYNC_SUBMIT=0: the exact-size path with the wait).
    int async_submit = 1;
    // ORB stages of consecutive units take turns (each waits for the previous unit's ORB stage on the GPU, event to event):
    // what the host wait used to enforce as a side effect (SLIDEO_ORB_CHAIN=0: free-running).
    int orb_chain = 1;
    hipEvent_t last_orb_ev = nullptr;

    // workspaces
    slideo::Slot slots[slideo::NSLOTS];
    int next_slot = 0;
    int64_t next_ticket = 1;
    slideo::DevBuf d_small, d_ssd, d_prev_small, d_tapq, d_tapt, d_tapidx, d_tapdist;
    struct SiftWs { slideo::DevBuf base, gauss, gray, cand, counts, raw, items, kept, qofs, info, kp, desc; } sift;     // csrc/sift.hip.h

    // stage profiling (HIP events on the launch streams)
    bool profiling = false;
    double prof_ms[SLIDEO_N_STAGES] = {0, 0, 0, 0};
    int64_t prof_n[SLIDEO_N_STAGES] = {0, 0, 0, 0};
    int64_t prof_pairs = 0;
    slideo::DevBuf d_clk;           // {shader cycles, 100 MHz ticks, samples} summed by the search blocks while profiling (knn_tile.hip.h KtClock)

    // trace of the last match call
    std::vector<slideo::FrameCands> last_fcs;
};

namespace slideo {

// ---- capi_runtime.hip ---------------------------------------------------------------------------
void set_err(slideo_matcher* m, const char* what);
void check_launch(const char* what);
GeomEntry& geom_for(slideo_matcher* m, int w, int h);
int area_class_for(slideo_matcher* m, int w, int h);
void upload_area(slideo_matcher* m);
inline bool blur_is_f32(const slideo_matcher* m) { return m->cfg.ocv.blur <= 1; }
uint32_t kp_cap_for(const slideo_matcher* m, const PyrGeom& g);
int sub_batch_for(slideo_matcher* m, const PyrGeom& g, int n);
void require_idle(slideo_matcher* m);
// slot 0's staging buffer with room for `bytes` (taps, page ingest, the changed mask): whatever slideo_changed_mask_bgr8 kept there
// is gone afterwards
uint8_t* stage_for_upload(slideo_matcher* m, size_t bytes);
void upload_frames(Slot& S, const uint8_t* host, int n, int h, int stride, int64_t frame_stride, hipStream_t cs = nullptr);
bool host_is_pinned(const void* p);
void validate_image(int w, int h, int stride);
void upload_rng_stream(slideo_matcher* m, uint32_t len);
void unit_submit(slideo_matcher* m, Slot& S, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride, bool allow_async = true);
void unit_collect(slideo_matcher* m, Slot& S, slideo_verdict* out_host);
void check_match_args(slideo_matcher* m, int n, const void* frames, const void* out, int w, int h, int stride, int64_t frame_stride);
void match_frames_impl(slideo_matcher* m, int n, const uint8_t* frames, bool on_device, int w, int h, int stride, int64_t frame_stride,
                       slideo_verdict* out, hipStream_t user_stream);
// ProcessedImage::compute over n host pages (mo/lib.rs:92-131) WITHOUT appending them: the analysed pages, in order, into `out`
void analyse_pages(slideo_matcher* m, int n_pages, const uint8_t* const* data, const int32_t* width, const int32_t* height, const int32_t* stride_bytes,
                   std::vector<HostPage>& out, uint64_t progress_base, uint64_t progress_total);
// appends an analysed page (this matcher's own, or another device's of the same config: the records are plain host data)
void append_page(slideo_matcher* m, const HostPage& pg);

// ---- stage_orb.hip --------------------------------------------------------------------------------
void orb_stage_init(slideo_matcher* m);          // device tables of the ORB kernels + their launch attributes (slideo_matcher_create)
void orb_geom_init(slideo_matcher* m, GeomEntry& e, const std::vector<uint32_t>& lin_tab);      // per frame size: the kernels' device tables
void orb_stage1(slideo_matcher* m, Slot& S, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride,
                bool with_blur = false, uint32_t kp_cap = 0xFFFFFFFFu);
void orb_wait_info(slideo_matcher* m, Slot& S);
void orb_stage2(slideo_matcher* m, Slot& S, int w, int h, bool by_capacity = false);
void run_orb(slideo_matcher* m, Slot& S, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride,
             bool keep_host_qofs, bool with_blur = false);
// the two ORB kernels the SIFT stage shares: BGR -> gray u8 (pitch `pitch`, frame stride gframe), and the per-frame offsets scan
void orb_launch_gray(const slideo_matcher* m, const uint8_t* frames_dev, int64_t frame_stride, int stride, uint8_t* gray, int64_t gframe, int w, int h,
                     int pitch, int n, hipStream_t st);
void orb_launch_scan(const uint32_t* counts, int n, uint32_t* qofs, uint32_t* info, hipStream_t st);

// ---- stage_knn.hip --------------------------------------------------------------------------------
// FlannMatcher::new (mo/flann.rs:65-71) for the Hamming index: uploads the M packed rows, collapses equal rows, builds the
// matrix-core operand (and the LSH tables for matcher 1).  Sets m->Mu.
void knn_build_index(slideo_matcher* m, const std::vector<uint8_t>& train, int64_t M);
// workspace of a unit's search (before the timed interval) and the search itself: S.d_desc -> S.d_keys (+ the expansion of the
// collapsed rows).  qplan: the query count the launch is planned for, qtot: the capacity (async) or the real count.
void knn_reserve_unit(slideo_matcher* m, Slot& S, uint32_t qplan, uint32_t qtot);
void unit_knn(slideo_matcher* m, Slot& S, int n, uint32_t qplan, uint32_t qtot, bool async, bool prof, hipStream_t st);
bool knn_unit_is_valu(const slideo_matcher* m, int nq);
int knn_unit_rows(const slideo_matcher* m, int nq);        // train rows a unit's search evaluates (Mu, or M for the VALU engine)
void l2_prepare(slideo_matcher::L2Set& L, const uint8_t* t, int nt, hipStream_t st);
void l2_query(slideo_matcher* m, slideo_matcher::L2Set& L, const uint8_t* q_dev, int nq, int k, hipStream_t st, Slot& S, bool timed,
              DevBuf* keys = nullptr, DevBuf* pend = nullptr, float prune_tol = 0.f);
// SIFT matcher mode: the vote rule applied to the L2 lists, as Hamming-format lists in S.d_keys
void l2_lists_to_keys(slideo_matcher* m, Slot& S, const DevBuf& lists, int kq, uint32_t qtot, bool lowe, hipStream_t st);

// ---- stage_verify.hip -----------------------------------------------------------------------------
void verify_stage_init(slideo_matcher* m);
VerifyParams make_vp(const slideo_config& c);
void unit_verify(slideo_matcher* m, Slot& S, const VerifyParams& vp, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride,
                 uint32_t qtot);
void run_small(slideo_matcher* m, const uint8_t* imgs_dev, int n, int w, int h, int stride, int64_t img_stride, int& sw, int& sh, hipStream_t st);
// ssd[i] = sum of squared differences of the small images a + i * a_stride and b + i * b_stride (`bytes` each), i < n
void launch_ssd(const uint8_t* a, int64_t a_stride, const uint8_t* b, int64_t b_stride, int64_t bytes, unsigned long long* ssd, int n, hipStream_t st);

// ---- stage_sift.hip -------------------------------------------------------------------------------
void sift_check_cfg(const slideo_sift_config* sc, int w, int h);
void unit_submit_sift(slideo_matcher* m, Slot& S, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride);
void add_pages_sift(slideo_matcher* m, Slot& S, int cnt, int w, int h, int stride, int64_t fb);

}  // namespace slideo

#define API_TRY try {
#define API_CATCH(m)                                                          \
    }                                                                         \
    catch (const slideo::Error& e) { slideo::set_err(m, e.what()); return e.code; }   \
    catch (const std::exception& e) { slideo::set_err(m, e.what()); return SLIDEO_ERR_HIP; } \
    catch (...) { slideo::set_err(m, "unknown error"); return SLIDEO_ERR_HIP; }       \
    return SLIDEO_OK;
