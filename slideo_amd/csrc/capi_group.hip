// capi_group.hip — slideo_group_*: one matcher per device behind ONE handle (include/slideo_amd.h, "N-device group").
//
// The reference fans a video's frames out over every core of the machine (rayon: one task per changed frame,
// crates/matching-opencv/src/lib.rs:174,213) and its pages over the same pool (lib.rs:45-47).  The N-device counterpart: the
// page database is REPLICATED on every device (0.45 GB at 1000 pages against 288 GB of HBM), a call's
// frames are cut into contiguous shards, one per device, each shard runs through its device's matcher on a host thread of its
// own, and every shard's verdict records land in the caller's array at the shard's offset — the gather of the in-process
// form is the D2H copy each device makes anyway, so there is no collective here (one process per GPU, where every rank wants
// the whole timeline, is bench.py / slideo_amd/distributed.py: ONE RCCL all-gather of the 16-byte records).
// Page analysis is sharded the same way (ProcessedImage::compute is independent per page): device d analyses its share of a
// call's pages and every member appends the whole call, in page order, from the host records.
//
// No kernel lives in this unit; it only drives the members through the functions of runtime.hpp.
#include "runtime.hpp"

#include <thread>

using namespace slideo;

struct slideo_group {
    std::vector<slideo_matcher*> members;
    std::string err;
    slideo_progress_fn progress = nullptr;
    void* progress_user = nullptr;
    // progress of a sharded call: the members report their own (done, total); the group reports the sum
    struct Tramp { slideo_group* g = nullptr; uint64_t last = 0; };
    std::vector<Tramp> tramps;
    // (reports are serialised: one member thread at a time advances `done` and calls the sink, so a sink sees a monotonic count
    // and is never entered twice at once — what a single matcher's caller gets)
    std::mutex progress_mutex;
    uint64_t done = 0;
    uint64_t total = 0;
    const char* msg_override = nullptr;
    // shards of the last match call (trace lookup) and of the last changed-mask call (kept frames)
    std::vector<int> match_lo;           // n + 1 bounds
    struct KeptShard { int read_lo = 0, lo = 0, hi = 0; };
    std::vector<KeptShard> kept;
    bool kept_valid = false;
};

namespace {

std::string g_group_create_error;
std::mutex g_group_err_mutex;

void set_group_err(slideo_group* g, const char* what) {
    if (g) g->err = what;
    else { std::lock_guard<std::mutex> lk(g_group_err_mutex); g_group_create_error = what; }
}

// contiguous block [lo, hi) of member r (block sizes differ by at most one): slideo_amd/distributed.py shard_range
void shard_range(int n, int r, int world, int& lo, int& hi) {
    const int base = n / world, rem = n % world;
    lo = r * base + std::min(r, rem);
    hi = lo + base + (r < rem ? 1 : 0);
}

void group_progress_tramp(void* user, uint64_t done, uint64_t total, const char* msg) {
    auto* t = static_cast<slideo_group::Tramp*>(user);
    slideo_group* g = t->g;
    (void)total;
    if (!g->progress || done <= t->last) { t->last = std::max(t->last, done); return; }
    const uint64_t d = done - t->last;
    t->last = done;
    std::lock_guard<std::mutex> lk(g->progress_mutex);
    g->done += d;
    g->progress(g->progress_user, std::min(g->done, g->total), g->total, g->msg_override ? g->msg_override : msg);
}

// fn(member index) on one host thread per member; the first failure (lowest member) is rethrown on the caller's thread
template <class F>
void for_each_member(slideo_group* g, F fn) {
    const int n = (int)g->members.size();
    std::vector<int32_t> code((size_t)n, SLIDEO_OK);
    std::vector<std::string> what((size_t)n);
    auto body = [&](int r) {
        try { fn(r); }
        catch (const slideo::Error& e) { code[r] = e.code; what[r] = e.what(); }
        catch (const std::exception& e) { code[r] = SLIDEO_ERR_HIP; what[r] = e.what(); }
        catch (...) { code[r] = SLIDEO_ERR_HIP; what[r] = "unknown error"; }
    };
    if (n == 1) body(0);
    else {
        std::vector<std::thread> th;
        th.reserve((size_t)n);
        try {
            for (int r = 0; r < n; ++r) th.emplace_back(body, r);
        } catch (...) {                                     // (a thread could not be started: the members it would have driven run here)
            for (int r = (int)th.size(); r < n; ++r) body(r);
        }
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < n; ++r)
        if (code[r] != SLIDEO_OK) fail(code[r], "device member %d (device %d): %s", r, g->members[r]->device, what[r].c_str());
}

void check_member_call(slideo_matcher* m, int32_t rc) {
    if (rc != SLIDEO_OK) fail(rc, "%s", slideo_last_error(m));
}

void begin_progress(slideo_group* g, uint64_t total, const char* msg_override) {
    g->done = 0; g->total = total; g->msg_override = msg_override;
    for (auto& t : g->tramps) t.last = 0;
}

}  // namespace

#define GROUP_TRY try {
#define GROUP_CATCH(g)                                                        \
    }                                                                         \
    catch (const slideo::Error& e) { set_group_err(g, e.what()); return e.code; }   \
    catch (const std::exception& e) { set_group_err(g, e.what()); return SLIDEO_ERR_HIP; } \
    catch (...) { set_group_err(g, "unknown error"); return SLIDEO_ERR_HIP; } \
    return SLIDEO_OK;

extern "C" {

int32_t slideo_device_list(int32_t* ordinals_out, int32_t capacity) {
    int ndev = 0, n = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    for (int d = 0; d < ndev; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::string(prop.gcnArchName).find("gfx950") != std::string::npos) {
            if (ordinals_out && n < capacity) ordinals_out[n] = d;
            ++n;
        }
    }
    return n;
}

int32_t slideo_device_count(void) { return slideo_device_list(nullptr, 0); }

int32_t slideo_group_create(const slideo_config* cfg, int32_t n_devices, const int32_t* devices, slideo_group** out) {
    slideo_group* none = nullptr;
    GROUP_TRY
    if (!cfg || !out) fail(SLIDEO_ERR_INVALID_ARG, "null cfg/out");
    *out = nullptr;
    // n_devices 0 (devices may be null): every gfx950 device of the node, by its own HIP ordinal — a device of another
    // architecture at a lower ordinal shifts nothing (the count alone would name ordinals 0 .. n-1)
    std::vector<int32_t> all;
    if (n_devices == 0) {
        all.resize(64);
        const int32_t n = slideo_device_list(all.data(), 64);
        if (n < 1) fail(SLIDEO_ERR_NO_DEVICE, "no gfx950 device visible (this library has no CPU fallback)");
        all.resize((size_t)std::min(n, 64));
        devices = all.data(); n_devices = (int32_t)all.size();
    }
    if (!devices) fail(SLIDEO_ERR_INVALID_ARG, "null devices with n_devices > 0");
    if (n_devices < 1 || n_devices > 64) fail(SLIDEO_ERR_INVALID_ARG, "n_devices must be 0 (all) or 1..64");
    std::unique_ptr<slideo_group> g(new slideo_group());
    struct Guard { slideo_group* g; ~Guard() { if (g) for (slideo_matcher* m : g->members) slideo_matcher_destroy(m); } } guard{g.get()};
    for (int i = 0; i < n_devices; ++i) {
        slideo_matcher* m = nullptr;
        const int32_t rc = slideo_matcher_create(cfg, devices[i], &m);
        if (rc != SLIDEO_OK) fail(rc, "member %d (device %d): %s", i, devices[i], slideo_last_error(nullptr));
        g->members.push_back(m);
    }
    g->tramps.resize((size_t)n_devices);
    for (auto& t : g->tramps) t.g = g.get();
    guard.g = nullptr;
    *out = g.release();
    GROUP_CATCH(none)
}

void slideo_group_destroy(slideo_group* g) {
    if (!g) return;
    for (slideo_matcher* m : g->members) slideo_matcher_destroy(m);
    delete g;
}

const char* slideo_group_last_error(const slideo_group* g) {
    if (g) return g->err.c_str();
    std::lock_guard<std::mutex> lk(g_group_err_mutex);
    static thread_local std::string copy;
    copy = g_group_create_error;
    return copy.c_str();
}

int32_t slideo_group_device_count(const slideo_group* g) { return g ? (int32_t)g->members.size() : 0; }

slideo_matcher* slideo_group_member(slideo_group* g, int32_t i) {
    return (g && i >= 0 && i < (int)g->members.size()) ? g->members[i] : nullptr;
}

int32_t slideo_group_set_progress(slideo_group* g, slideo_progress_fn fn, void* user) {
    if (!g) return SLIDEO_ERR_INVALID_ARG;
    g->progress = fn; g->progress_user = user;
    for (size_t r = 0; r < g->members.size(); ++r)
        slideo_matcher_set_progress(g->members[r], fn ? group_progress_tramp : nullptr, fn ? &g->tramps[r] : nullptr);
    return SLIDEO_OK;
}

int32_t slideo_group_use_sift(slideo_group* g, const slideo_sift_config* cfg, float ratio) {
    if (!g) return SLIDEO_ERR_INVALID_ARG;
    GROUP_TRY
    for (slideo_matcher* m : g->members) check_member_call(m, slideo_matcher_use_sift(m, cfg, ratio));
    GROUP_CATCH(g)
}

int32_t slideo_group_add_pages_bgr8(slideo_group* g, int32_t n_pages, const uint8_t* const* data, const int32_t* width,
                                    const int32_t* height, const int32_t* stride_bytes) {
    if (!g) return SLIDEO_ERR_INVALID_ARG;
    GROUP_TRY
    if (n_pages < 0 || (n_pages > 0 && (!data || !width || !height || !stride_bytes))) fail(SLIDEO_ERR_INVALID_ARG, "null page arrays");
    for (slideo_matcher* m : g->members) if (m->finalized) fail(SLIDEO_ERR_STATE, "pages cannot be added after finalize");
    const int N = (int)g->members.size();
    const uint64_t total = (uint64_t)n_pages;
    begin_progress(g, total, nullptr);
    if (g->progress) g->progress(g->progress_user, 0, total, "Analyzing PDF pages...");              // lib.rs:43
    // every member analyses its contiguous share of the call's pages (lib.rs:45-47: independent per page) ...
    std::vector<std::vector<HostPage>> got((size_t)N);
    for_each_member(g, [&](int r) {
        int lo, hi;
        shard_range(n_pages, r, N, lo, hi);
        if (hi > lo) analyse_pages(g->members[r], hi - lo, data + lo, width + lo, height + lo, stride_bytes + lo, got[r], 0, (uint64_t)(hi - lo));
    });
    // ... and every member's database receives the whole call, in page order — each on its own thread (the records are plain
    // host data: N copies of the deck's records side by side instead of one after the other on the caller's thread)
    for_each_member(g, [&](int me) {
        for (int r = 0; r < N; ++r)
            for (const HostPage& pg : got[r]) append_page(g->members[me], pg);
    });
    if (g->progress) g->progress(g->progress_user, total, total, "PDF page analysis successful.");   // lib.rs:58
    GROUP_CATCH(g)
}

int32_t slideo_group_finalize_pages(slideo_group* g) {
    if (!g) return SLIDEO_ERR_INVALID_ARG;
    GROUP_TRY
    for_each_member(g, [&](int r) { check_member_call(g->members[r], slideo_matcher_finalize_pages(g->members[r])); });
    GROUP_CATCH(g)
}

int32_t slideo_group_page_count(const slideo_group* g) { return g ? slideo_matcher_page_count(g->members[0]) : -1; }
int64_t slideo_group_descriptor_count(const slideo_group* g) { return g ? slideo_matcher_descriptor_count(g->members[0]) : -1; }

int32_t slideo_group_match_frames_bgr8(slideo_group* g, int32_t n_frames, const uint8_t* frames, int32_t width, int32_t height,
                                       int32_t stride_bytes, int64_t frame_stride_bytes, slideo_verdict* verdicts_out) {
    if (!g) return SLIDEO_ERR_INVALID_ARG;
    GROUP_TRY
    if (n_frames < 0 || (n_frames > 0 && (!frames || !verdicts_out))) fail(SLIDEO_ERR_INVALID_ARG, "null frames/verdicts");
    const int N = (int)g->members.size();
    begin_progress(g, (uint64_t)n_frames, nullptr);
    g->match_lo.assign((size_t)N + 1, 0);
    for (int r = 0; r < N; ++r) { int lo, hi; shard_range(n_frames, r, N, lo, hi); g->match_lo[r] = lo; g->match_lo[r + 1] = hi; }
    g->kept_valid = false;
    for_each_member(g, [&](int r) {
        const int lo = g->match_lo[r], hi = g->match_lo[r + 1];
        // (an empty shard still runs the call's checks: every member reports a matcher that was never finalized, say)
        match_frames_impl(g->members[r], hi - lo, frames + (int64_t)lo * frame_stride_bytes, false, width, height, stride_bytes, frame_stride_bytes,
                          verdicts_out + lo, nullptr);
    });
    GROUP_CATCH(g)
}

int32_t slideo_group_last_frame_candidates(const slideo_group* g, int32_t frame_in_batch, slideo_candidate* out, int32_t capacity, int32_t* n_out) {
    if (!g || !n_out || g->match_lo.empty()) return SLIDEO_ERR_INVALID_ARG;
    for (size_t r = 0; r + 1 < g->match_lo.size(); ++r)
        if (frame_in_batch >= g->match_lo[r] && frame_in_batch < g->match_lo[r + 1])
            return slideo_last_frame_candidates(g->members[r], frame_in_batch - g->match_lo[r], out, capacity, n_out);
    return SLIDEO_ERR_INVALID_ARG;
}

int32_t slideo_group_changed_mask_bgr8(slideo_group* g, int32_t n_frames, const uint8_t* frames, int32_t width, int32_t height,
                                       int32_t stride_bytes, int64_t frame_stride_bytes, const uint8_t* prev_small,
                                       uint8_t* last_small_out, uint8_t* changed_out, float* similarity_out) {
    if (!g) return SLIDEO_ERR_INVALID_ARG;
    GROUP_TRY
    if (n_frames < 0 || (n_frames > 0 && (!frames || !changed_out))) fail(SLIDEO_ERR_INVALID_ARG, "null frames/changed");
    g->kept_valid = false;
    if (n_frames == 0) return SLIDEO_OK;
    const int N = (int)g->members.size();
    // MarkSimilarIter compares every sampled frame with the one before it (video_capture.rs:86-98): a shard reads ONE frame
    // before its block (the halo; slideo_amd/distributed.py halo_range) and drops that frame's own flag.
    g->kept.assign((size_t)N, slideo_group::KeptShard{});
    int last_r = 0;
    for (int r = 0; r < N; ++r) {
        int lo, hi;
        shard_range(n_frames, r, N, lo, hi);
        g->kept[r] = slideo_group::KeptShard{(lo > 0 && hi > lo) ? lo - 1 : lo, lo, hi};
        if (hi > lo) last_r = r;
    }
    for_each_member(g, [&](int r) {
        const slideo_group::KeptShard k = g->kept[r];
        if (k.hi <= k.lo) return;
        const int cnt = k.hi - k.read_lo, halo = k.lo - k.read_lo;
        std::vector<uint8_t> ch((size_t)cnt);
        std::vector<float> sim((size_t)cnt);
        check_member_call(g->members[r], slideo_changed_mask_bgr8(g->members[r], cnt, frames + (int64_t)k.read_lo * frame_stride_bytes, width, height, stride_bytes,
                                                                   frame_stride_bytes, r == 0 ? prev_small : nullptr,
                                                                   r == last_r ? last_small_out : nullptr, ch.data(), sim.data()));
        for (int i = halo; i < cnt; ++i) {
            changed_out[k.read_lo + i] = ch[i];
            if (similarity_out) similarity_out[k.read_lo + i] = sim[i];
        }
    });
    g->kept_valid = true;
    GROUP_CATCH(g)
}

int32_t slideo_group_match_kept_frames(slideo_group* g, int32_t n_sel, const int32_t* sel, slideo_verdict* verdicts_out) {
    if (!g) return SLIDEO_ERR_INVALID_ARG;
    GROUP_TRY
    if (n_sel < 0 || (n_sel > 0 && (!sel || !verdicts_out))) fail(SLIDEO_ERR_INVALID_ARG, "null selection/verdicts");
    if (!g->kept_valid) fail(SLIDEO_ERR_STATE, "no frames kept: slideo_group_changed_mask_bgr8 must be the call before (its upload is what is matched)");
    const int N = (int)g->members.size();
    const int n_kept = g->kept.empty() ? 0 : g->kept.back().hi;
    // a selected frame is matched by the member whose block holds it, from the copy the mask call left on that device
    std::vector<std::vector<int32_t>> local((size_t)N), where((size_t)N);
    for (int i = 0; i < n_sel; ++i) {
        if (sel[i] < 0 || sel[i] >= n_kept) fail(SLIDEO_ERR_INVALID_ARG, "selected frame %d outside the %d kept", sel[i], n_kept);
        for (int r = 0; r < N; ++r)
            if (sel[i] >= g->kept[r].lo && sel[i] < g->kept[r].hi) { local[r].push_back(sel[i] - g->kept[r].read_lo); where[r].push_back(i); break; }
    }
    begin_progress(g, (uint64_t)n_sel, nullptr);
    g->match_lo.clear();
    for_each_member(g, [&](int r) {
        if (local[r].empty()) return;
        std::vector<slideo_verdict> v(local[r].size());
        check_member_call(g->members[r], slideo_match_kept_frames(g->members[r], (int32_t)local[r].size(), local[r].data(), v.data()));
        for (size_t j = 0; j < v.size(); ++j) verdicts_out[where[r][j]] = v[j];
    });
    GROUP_CATCH(g)
}

}  // extern "C"
