// types.h — plain-data types and constants shared by the kernels (csrc/*.hip.h) and the host runtime (runtime.hpp).
// No kernel lives here: every kernel header is compiled by exactly ONE translation unit (stage_*.hip), and whatever the
// other units need of a stage — its records, its key format — is in this file.
#pragma once
#include <stdint.h>

#include "geom.h"

namespace slideo {

// ---- k-NN key format (knn.hip.h): key = distance << 23 | train_row ----
constexpr int KNN_KEY_SHIFT = 23;
constexpr uint32_t KNN_IDX_MASK = (1u << KNN_KEY_SHIFT) - 1;
constexpr uint32_t KNN_EMPTY = 0xFFFFFFFFu;
constexpr int KNN_BLOCK = 256;

// ---- ORB tables (orb.hip.h) ----
struct OrbTables {             // device-resident constants (per matcher)
    int32_t umax[68];
    int32_t gk[8];             // 7-tap Q8 kernel of slideo_ocv_variants.blur 2 (sum 257) / 3 (sum 256); geom.h
    float gkf[8];              // 7-tap f32 kernel of blur 0 / 1
    int8_t pattern[1024];      // 512 (x,y)
};

struct GrayCoef { uint32_t cb, cg, cr, shift; };      // slideo_ocv_variants.gray: Q15 3735/19235/9798 or Q14 1868/9617/4899

// ---- verification records (verify.hip.h, homography.hip.h) ----
constexpr int VOTE_KLIST = 32;   // key-list stride of the kNN stage (KLIST in slideo_capi.hip; checked there)
constexpr int MAXC = 64;       // >= max_candidate_pages
constexpr int MAXR = 16;       // >= max_rated
constexpr int RANSAC_SMALL_PTS = 256;   // candidates with at most this many votes go to the small-LDS instance
constexpr int RANSAC_LDS_PTS = 1024;    // point pairs kept in LDS (more go through global memory); 17 KB per 64-thread block = 9 blocks per CU

struct FrameCands {            // one per frame of the batch, device resident
    int32_t ncand, nsurv;
    int32_t page[MAXC], count[MAXC], ofs[MAXC];
    int32_t inliers[MAXC], found[MAXC];
    double M[MAXC][9];         // slide -> frame: 2x3 (verify_model 0, entries 6-8 unused) or 3x3 homography (verify_model 1)
    int32_t surv[MAXR];        // candidate slot of each survivor
    float sim[MAXR];
    unsigned long long ssd[MAXR];
};

struct PageInfo {              // per page, device resident
    int32_t w, h;              // full size
    int32_t area_idx;          // AreaGeom index (size class)
    int32_t sw, sh;            // small size
    int32_t kp_ofs, kp_cnt;    // rows of this page in the train matrix
    int32_t _pad;
    int64_t small_ofs;         // byte offset of the small image
};

struct PairDesc {              // one (frame, survivor) unit of re-projection work, written by rate_kernel
    int32_t f, s, area_idx, _pad;
    int64_t small_ofs;
    double M[9];
};

struct VerifyParams {
    int32_t k, klist, max_cand, max_rated;
    float tol, min_similarity, ratio;       // ratio > 0: ratio test instead of the tolerance vote
    double thr, conf, min_rating, min_rating_ratio;
    int32_t max_iters, refine_iters;
    uint32_t rng_len;                       // entries of the pre-drawn cv::RNG stream (grown on demand by the host)
    int32_t model;                          // slideo_config.verify_model: 0 similarity (2x3), 1 homography (3x3)
    int32_t verdict_rule;                   // slideo_config.verdict_rule: 0 best similarity wins (mo/lib.rs:370-389), 1 rating order, similarity only accepts
    int32_t sched_window;                   // ransac_kernel: redraw schedule from the LDS-window jump tables (1) / by the fixed point only (0: A/B, tests)
};

// ---- LSH index on the device (knn_lsh.hip.h) ----
struct LshDev {
    LshParams p;
    int32_t nbuckets;                   // 2^kb
    const int32_t* ofs;                 // [ntab][nbuckets + 1]
    const int32_t* rows;                // [ntab][M]
    const uint16_t* keys;               // [M][ntab]
    int32_t M;
};

}  // namespace slideo
