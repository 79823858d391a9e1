// stage_verify.hip — drivers of the verification stage, vote .. verdict, and of to_small_image (kernels: verify.hip.h, homography.hip.h).
#include "runtime.hpp"
#include "verify.hip.h"
#include "homography.hip.h"

using namespace slideo;

namespace slideo {

VerifyParams make_vp(const slideo_config& c) {
    VerifyParams v{};
    v.k = c.knn_k; v.klist = KLIST; v.max_cand = c.max_candidate_pages; v.max_rated = c.max_rated;
    v.tol = c.vote_tolerance; v.min_similarity = c.min_similarity; v.ratio = c.ratio_test;
    v.thr = c.ransac_threshold; v.conf = c.ransac_confidence; v.min_rating = c.min_rating;
    v.min_rating_ratio = c.min_rating_ratio; v.max_iters = c.ransac_max_iters; v.refine_iters = c.refine_iters;
    v.model = c.verify_model; v.verdict_rule = c.verdict_rule;
    v.sched_window = env_long("SLIDEO_RANSAC_WINDOW", 1) != 0;      // (read per unit: the tests switch it)
    return v;
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

void launch_ssd(const uint8_t* a, int64_t a_stride, const uint8_t* b, int64_t b_stride, int64_t bytes, unsigned long long* ssd, int n, hipStream_t st) {
    ssd_kernel<<<n, 256, 0, st>>>(a, a_stride, b, b_stride, bytes, ssd);
    check_launch("ssd_kernel");
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
            const long tail_env = env_long("SLIDEO_RH_TAIL_ROUNDS", 256);                     // (read per unit: the tests switch it)
            const uint32_t tail_rounds = (uint32_t)(tail_env < 1 ? 0xFFFFFFFFu : tail_env);   // 0 / negative: never hand over
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
                const int lane_lm = (int)env_long("SLIDEO_REFINE_LANE_LM", ncand >= 4096 ? 1 : 0);      // (read per unit: the tests switch it)
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
        // the pairs of size classes inside reproject_vt_kernel's limits go through the warped-image tile, the others through the
        // frame window of reproject_kernel (each kernel skips the other's pairs; the second launch only if such a class exists)
        int max_tile_rows = 0;
        bool any_vt = false, any_win = false;
        for (const AreaGeom& ag : m->area_geoms) {
            max_tile_rows = std::max(max_tile_rows, cdiv(ag.dh, SM_TH));
            const bool vt = ag.vt_ok && cdiv(ag.dw, SM_TW) <= RP_MAX_TILES;
            any_vt |= vt; any_win |= !vt;
        }
        const dim3 rgrid(max_tile_rows, std::min(n, 65535));
#define SLIDEO_RP_TAIL m->d_page_small.as<uint8_t>(), frames_dev, frame_stride, stride, w, h, S.d_fcs.as<FrameCands>(), pair_list, pair_count
#define SLIDEO_RP_ARGS m->d_area_geoms.as<AreaGeom>(), m->d_area_taps.as<AreaTap>(), m->d_area_idx.as<int32_t>(), SLIDEO_RP_TAIL
#define SLIDEO_VT_ARGS m->d_area_geoms.as<AreaGeom>(), m->d_area_taps.as<AreaTap>(), m->d_area_idx.as<int32_t>(), m->d_area_recs.as<AreaRec>(), SLIDEO_RP_TAIL
        if (any_vt) {
            if (c.verify_model == 1) reproject_vt_kernel<true><<<rgrid, 256, 0, st>>>(SLIDEO_VT_ARGS);
            else reproject_vt_kernel<false><<<rgrid, 256, 0, st>>>(SLIDEO_VT_ARGS);
            check_launch("reproject_vt_kernel");
        }
        if (any_win) {
            if (c.verify_model == 1) reproject_kernel<true><<<rgrid, 256, 0, st>>>(SLIDEO_RP_ARGS);
            else reproject_kernel<false><<<rgrid, 256, 0, st>>>(SLIDEO_RP_ARGS);
            check_launch("reproject_kernel");
        }
#undef SLIDEO_RP_ARGS
#undef SLIDEO_VT_ARGS
#undef SLIDEO_RP_TAIL
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

// launch attributes of the stage's kernels (slideo_matcher_create)
void verify_stage_init(slideo_matcher* m) {
    (void)m;
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&vote_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
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
}

}  // namespace slideo

