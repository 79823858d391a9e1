// stage_orb.hip — drivers of the ORB stage (kernels: orb.hip.h).
#include "runtime.hpp"
#include "orb.hip.h"

using namespace slideo;

namespace slideo {

static GrayCoef gray_coef(const slideo_matcher* m) {
    return m->cfg.ocv.gray == 1 ? GrayCoef{1868u, 9617u, 4899u, 14u} : GrayCoef{3735u, 19235u, 9798u, 15u};
}

// device tables of the ORB kernels (umax, Gaussian taps, BRIEF pattern, intensity-centroid weights) and their launch attributes
void orb_stage_init(slideo_matcher* m) {
    const slideo_config* cfg = &m->cfg;
    OrbTables t{};
    umax_table(cfg->patch_size / 2, t.umax);
    if (cfg->ocv.blur == 2) gauss7_q8_rounded(t.gk); else gauss7_fixed(t.gk);
    gauss7_f32(t.gkf);
    brief_pattern(cfg->patch_size, t.pattern, cfg->ocv.rng_mul);
    m->d_tables.reserve(sizeof(OrbTables));
    HIP_CHECK(hipMemcpy(m->d_tables.p, &t, sizeof(t), hipMemcpyHostToDevice));
    std::vector<uint32_t> ict;
    ic_weight_table(cfg->patch_size / 2, t.umax, ict, m->ic_shift);
    m->ic_entries = (int)(ict.size() / 2);
    m->d_ictab.reserve(ict.size() * 4);
    HIP_CHECK(hipMemcpy(m->d_ictab.p, ict.data(), ict.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&describe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  describe_window(cfg->patch_size / 2).dwords * 16));
}

// per frame size: the resize tap table and the FAST tile table
void orb_geom_init(slideo_matcher* m, GeomEntry& e, const std::vector<uint32_t>& lin_tab) {
    std::vector<uint32_t> tab = lin_tab;
    if (tab.empty()) tab.push_back(0);
    e.lin_tab.reserve(tab.size() * 4);
    HIP_CHECK(hipMemcpyAsync(e.lin_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, m->stream));
    std::vector<int4> ft((size_t)std::max(e.g.fast_tiles, 1));
    for (int t = 0; t < e.g.fast_tiles; ++t) ft[t] = fast_tile_entry(e.g, t);
    e.fast_tiles.reserve(ft.size() * sizeof(int4));
    HIP_CHECK(hipMemcpyAsync(e.fast_tiles.p, ft.data(), ft.size() * sizeof(int4), hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));              // the host vectors die here
}

void orb_launch_gray(const slideo_matcher* m, const uint8_t* frames_dev, int64_t frame_stride, int stride, uint8_t* gray, int64_t gframe, int w, int h,
                     int pitch, int n, hipStream_t st) {
    const int aligned4 = ((uintptr_t)frames_dev % 4 == 0) && (stride % 4 == 0) && (frame_stride % 4 == 0);
    gray_kernel<<<dim3(cdiv(cdiv(w, 4), 256), h, n), 256, 0, st>>>(frames_dev, frame_stride, stride, gray, gframe, w, h, pitch, aligned4, gray_coef(m));
    check_launch("gray_kernel");
}

void orb_launch_scan(const uint32_t* counts, int n, uint32_t* qofs, uint32_t* info, hipStream_t st) {
    scan_kernel<<<1, 1024, 0, st>>>(counts, n, qofs, info);
    check_launch("scan_kernel");
}

// ---- ORB over `n` equally sized frames already on the device, in three steps -------------
// stage 1: gray, pyramid, FAST+NMS, blur, retainBest thresholds, per-frame offsets; copies {Qtot, max, flags} to pinned memory
static void launch_blur(slideo_matcher* m, Slot& S, const PyrGeom& g, int n, const uint8_t* strip_mask) {
    hipStream_t st = S.st;
    if (m->cfg.ocv.blur == 0)
        blur_f32_kernel<true><<<dim3(g.blur_tiles, n), 256, 0, st>>>(g, S.d_pyr.as<uint8_t>(), S.d_blur.as<uint8_t>(), m->d_tables.as<OrbTables>(), strip_mask);
    else if (m->cfg.ocv.blur == 1)
        blur_f32_kernel<false><<<dim3(g.blur_tiles, n), 256, 0, st>>>(g, S.d_pyr.as<uint8_t>(), S.d_blur.as<uint8_t>(), m->d_tables.as<OrbTables>(), strip_mask);
    else
        blur_kernel<<<dim3(g.blur_tiles, n), 256, 0, st>>>(g, S.d_pyr.as<uint8_t>(), S.d_blur.as<uint8_t>(), m->d_tables.as<OrbTables>());
    check_launch("blur kernel");
}

// SLIDEO_RESIZE_GENERIC=1: every pyramid level through resize_kernel (the form for any shrink factor) instead of resize_quad_kernel
static bool resize_generic_forced() { return env_long("SLIDEO_RESIZE_GENERIC", 0) != 0; }      // (read per unit: the tests switch it)

// `with_blur`: also materialise the WHOLE blurred pyramid (only the pyramid tap wants it)
// the f32 blur of ocv.blur 0 / 1 cannot be evaluated per BRIEF sample in integer arithmetic: those variants always
// materialise the blurred pyramid (blur_f32_kernel) and describe from it (describe_blurred_kernel)
void orb_stage1(slideo_matcher* m, Slot& S, const uint8_t* frames_dev, int n, int w, int h, int stride, int64_t frame_stride,
                bool with_blur, uint32_t kp_cap) {
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
    {
        dim3 grid(cdiv(cdiv(w, 4), 256), h, n);
        gray_kernel<<<grid, 256, 0, st>>>(frames_dev, frame_stride, stride, S.d_pyr.as<uint8_t>(), g.frame_bytes, w, h,
                                          g.lv[0].pitch, aligned4, gray_coef(m));
        check_launch("gray_kernel");
    }
    for (int l = 1; l < L; ++l) {
        if (g.lv[l].w <= 0 || g.lv[l].h <= 0) continue;
        // flat thread index t -> (row, 4-pixel group) = (t / nxq, t % nxq); magic = ceil(2^32 / nxq) divides
        // exactly while nxq^2 * h < 2^32, which MAX_DIM guarantees
        const int nxq = cdiv(g.lv[l].w, 4);
        const uint32_t magic = nxq > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)nxq - 1) / (uint64_t)nxq) : 0u;
        dim3 grid(cdiv(nxq * cdiv(g.lv[l].h, RESIZE_ROWS), 256), 1, n);
        if (g.lv[l].rq_ok && !resize_generic_forced())
            resize_quad_kernel<<<grid, 256, 0, st>>>(S.d_pyr.as<uint8_t>(), g.frame_bytes, g.lv[l - 1], g.lv[l], ge.lin_tab.as<uint32_t>(),
                                                     nxq, magic);
        else
            resize_kernel<<<grid, 256, 0, st>>>(S.d_pyr.as<uint8_t>(), g.frame_bytes, g.lv[l - 1], g.lv[l], ge.lin_tab.as<uint32_t>(),
                                                nxq, magic);
        check_launch("resize kernel");
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
void orb_stage2(slideo_matcher* m, Slot& S, int w, int h, bool by_capacity) {
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
             bool keep_host_qofs, bool with_blur) {
    orb_stage1(m, S, frames_dev, n, w, h, stride, frame_stride, with_blur);
    orb_wait_info(m, S);
    orb_stage2(m, S, w, h);
    if (keep_host_qofs) {
        S.orb.qofs.resize(n + 1);
        HIP_CHECK(hipMemcpyAsync(S.orb.qofs.data(), S.d_qofs.p, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost, S.st));
        HIP_CHECK(hipStreamSynchronize(S.st));
    }
}

}  // namespace slideo
