/*
 * slideo_amd.h — C ABI of the MI355X-native slide <-> video-frame matcher.
 *
 * This is the drop-in boundary for the hot path of hediet/slideo's
 * crates/matching-opencv.  The reference has no C ABI of its own (its only FFI
 * is the `opencv` crate's generated shims into OpenCV 4.5.2); each entry point
 * below names the reference call it replaces.  Paths are relative to the
 * reference repository root, shorthand `mo/` = crates/matching-opencv/src/.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every function returns int32_t status (SLIDEO_OK == 0); the message for
 *     the last failure on a handle is slideo_last_error(handle);
 *   - the reference surface has no Result anywhere (it panics, mo/lib.rs:95-104,
 *     mo/flann.rs:15-46); a binding mirrors that by panicking on status != 0;
 *   - no C++ exception crosses the ABI;
 *   - images are 8-bit, 3 channels, BGR interleaved, row-major with a byte
 *     stride (the cv::Mat 8UC3 layout the reference holds, mo/lib.rs:77-83);
 *   - "page" = one rasterised PDF page, "frame" = one decoded video frame,
 *     page indices are 0-based positions in the order pages were added.
 *   - there is NO CPU fallback: without a gfx950 device every compute entry
 *     point fails with SLIDEO_ERR_NO_DEVICE;
 *   - a matcher is NOT re-entrant: calls on one handle must come from one thread at
 *     a time (the reference's caller is single threaded, crates/app/src/main.rs:77-93,
 *     and parallelism lives inside the call); different handles are independent.
 *
 * Limits (each is checked; beyond it the call fails with SLIDEO_ERR_UNSUPPORTED and says which one — nothing is clamped):
 *   image width / height   1..4096        x and y of a FAST candidate travel packed in 12 bits each (geom.h MAX_DIM); the
 *                                         reference's inputs are <= 1920x1080 video frames and 2001x1125 renders, the 4K
 *                                         config is 3840x2160
 *   train descriptors      < 2^23         a k-NN key is distance << 23 | row; 500 pages x ~1850 = 0.92 M rows, 4000 pages fit
 *   pages                  <= 16384       the vote kernel keeps one counter per page in LDS
 *   knn_k                  1..32          the per-query list lives in registers (the reference uses 30)
 *   max_candidate_pages    1..64, max_rated 1..16, nlevels 1..16, ransac_max_iters 1..1000000
 *   lsh_tables 1..8, lsh_key_bits 1..16, lsh_multi_probe 0..2 (matcher 1)
 *   page / frame area      >= small_area  to_small_image (mo/image_utils.rs:8-20) only ever SHRINKS here; for an image below
 *                                         120 000 px OpenCV's INTER_AREA turns into a bilinear upscale, which is not restated
 *                                         (the reference's frames are >= 640x360)
 *   NOT limits: keypoints per frame (beyond 8192 the canonical sort moves from LDS to global memory; a frame beyond the
 *   capacity the asynchronous path provides for is re-run through the exact-size path), the RANSAC sample schedule (the
 *   pre-drawn cv::RNG stream is extended on demand), frames per call (cut into units that fit the workspace budget).
 *
 * Environment (read by slideo_matcher_create unless marked "per unit"; NONE changes a result — they select between code paths
 * that the tests hold bit-identical, or size workspaces; everything else that used to be switchable this way was removed):
 *   SLIDEO_KNN_ENGINE 0..3        initial value of slideo_matcher_set_knn_engine
 *   SLIDEO_KNN_SHARE=0 / 1        exact Hamming search: always two / always one block per CU (default: one while other units are in flight)
 *                      =3 / 4    the 12-wave block (three search waves per SIMD, one block per CU) while other units are in flight / always
 *   SLIDEO_KNN_W12_RATIO x        default rule: that 12-wave block while units share the chip when a unit carries >= x (query, train row) pairs per frame
 *                                 pixel (290: decks of ~750 pages x ORB-1000 and up at 1080p; 0 = never) — the fuller matrix pipe then outweighs the
 *                                 co-runners' occupancy (configs[3] + 6.8 %, configs[4] + 4.7 %; headline - 2..4 %, hence the rule)
 *                      =5 / 6    measurement: the 12-wave block of ONE query tile per wave (88 registers: three waves per SIMD in the registers of two,
 *                                csrc/knn_tile1.hip.h) while other units are in flight / always
 *   SLIDEO_KNN_NSEG n             measurement: train-stream segments of the matrix-core search instead of the plan's choice
 *   SLIDEO_KNN_DEDUP=0            search all M train rows instead of the distinct ones (slideo_matcher_unique_descriptor_count)
 *   SLIDEO_LSH_ENGINE=gather      matcher 1 by bucket gathering instead of the filtered matrix-core stream
 *   SLIDEO_ASYNC_SUBMIT=0         units through the exact-size path (one host wait for the keypoint counts in mid-unit)
 *   SLIDEO_ORB_CHAIN=0            ORB stages of consecutive units free-running instead of taking turns
 *   SLIDEO_STREAM_PICK=0          the slots' streams in plain creation order instead of picked by measurement to sit on hardware queues of their own (below)
 *   SLIDEO_RESIZE_GENERIC=1       every pyramid level through resize_kernel (any shrink factor) instead of resize_quad_kernel [per unit]
 *   SLIDEO_HOST_UNIT n            frames per unit of a host-memory batch (32; 0 = the device-path rule)
 *   SLIDEO_WS_GB x                workspace budget of all slots together (48)
 *   SLIDEO_SIFT_WS_MB n           SIFT pyramid budget per pass (98304 on a device with >= 192 GB, else 24576)  [per call]
 *   SLIDEO_SIFT_LIST_CAP n        start capacity of SIFT's per-frame extrema list (65536; the tests force its growth path) [per call]
 *   SLIDEO_RNG_STREAM_LEN n       start length of the pre-drawn cv::RNG stream (the tests force its growth path)
 *   SLIDEO_RANSAC_WINDOW=0        ransac_kernel's redraw schedule by the fixed point only                       [per unit]
 *   SLIDEO_RH_TAIL_ROUNDS n       verify_model 1: rounds before a candidate moves to ransac_h_tail_kernel (256; 0 = never) [per unit]
 *   SLIDEO_REFINE_LANE_LM 0|1     verify_model 1: the small candidates' LM in refine_h_kernel<1> / in the eigen kernel's lanes [per unit]
 *   (not the library's, but it decides how its streams run) GPU_MAX_HW_QUEUES — the HIP runtime maps a process's streams onto this many
 *       hardware queues (default 4) in creation order, and a matcher's four slot streams must not share one (their units' kernels would
 *       serialise: - 10 %, measured behind an initialised RCCL communicator, whose streams come first: profiles/r06_experiments.txt 6).
 *       slideo_matcher_create therefore PICKS its slot streams by measurement (a candidate is kept iff a kernel on it completes while spin
 *       kernels keep the streams kept so far busy; a few ms); with fewer hardware queues than slots it takes what there is.
 *   build time only: SLIDEO_HIP_EXTRA_FLAGS (slideo_amd/build.py), SLIDEO_LIB_PATH / SLIDEO_REBUILD (slideo_amd/_capi.py)
 */
#ifndef SLIDEO_AMD_H
#define SLIDEO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLIDEO_ABI_VERSION 7

enum {
    SLIDEO_OK = 0,
    SLIDEO_ERR_INVALID_ARG = 1,   /* null pointer, bad size, bad config        */
    SLIDEO_ERR_NO_DEVICE = 2,     /* no HIP device / not gfx950                */
    SLIDEO_ERR_HIP = 3,           /* a HIP runtime call failed                 */
    SLIDEO_ERR_STATE = 4,         /* call order violated (e.g. match before finalize) */
    SLIDEO_ERR_UNSUPPORTED = 5,   /* config outside what the kernels implement */
    SLIDEO_ERR_EMPTY_INDEX = 6,   /* no page produced a descriptor (reference: FLANN train on empty set throws, mo/flann.rs:45-47) */
    SLIDEO_ERR_CAPACITY = 7       /* caller-provided output buffer too small   */
};

/* OpenCV-semantics switches.  The arithmetic of the reference's hot path lives in OpenCV 4.5.2 C++
 * (Cargo.lock:1723-1724, .github/workflows/ci.yml:18), whose source is neither under the reference
 * repository nor in this image; every primitive whose exact rounding could only be RECALLED (SURVEY.md
 * Appendix A, confidence M / L) is therefore restated in more than one form, selected here, shared by the
 * kernels' host tables (csrc/geom.h) and by the CPU restatement (oracle/).  Value 0 is the default of every switch that
 * restates a call the reference makes = the best estimate of what a stock 4.5.2 build (SSE3 baseline, AVX2 dispatch, IPP on:
 * ci/install-bionic.sh) runs for it; DESIGN.md section 5 says why.  (`hdlt`, which belongs to an extension the reference never
 * runs, defaults to 1: see there.)  tools/pin_opencv.py dumps OpenCV's own
 * outputs where cv2 4.5.2 exists, and tests/test_opencv_pin.py then names the matching value of each switch.
 * The HIP library implements the values marked [hip]; others fail slideo_matcher_create with
 * SLIDEO_ERR_UNSUPPORTED (the CPU restatement implements all of them). */
typedef struct slideo_ocv_variants {
    /* cvtColor(BGR2GRAY) inside ORB::detectAndCompute, mo/feature_extractor.rs:32-40 [OCV A.1]
     *   0 [hip] Q15 coefficients 3735/19235/9798, (x + 2^14) >> 15   (4.x color_rgb.simd.hpp)
     *   1 [hip] Q14 coefficients 1868/9617/4899,  (x + 2^13) >> 14   (2.4 / 3.x)                      */
    int32_t gray;
    /* GaussianBlur(level, 7x7, sigma 2, BORDER_REFLECT_101) inside ORB, same call [OCV A.6].  ORB blurs a
     * SUBMATRIX of its pyramid buffer, which GaussianBlur's fixed-point branch excludes
     * (smooth.dispatch.cpp: `sdepth == CV_8U && ((borderType & BORDER_ISOLATED) || !src.isSubmatrix())`), so the
     * call falls through to sepFilter2D with the CV_32F kernel:
     *   0 [hip] sepFilter2D in f32 (4.2+: createBitExactKernel_32S rejects a kernel whose taps x 256 are not
     *           integers): row pass k0*p0 then += k_i*p_i, column pass k3*c then += k_j*(r_+j + r_-j),
     *           saturate_cast<uchar>(cvRound); products CONTRACTED to fma (the AVX2-dispatched filter.avx2.cpp
     *           of a stock build is compiled with -mfma and GCC's default -ffp-contract=fast)
     *   1 [hip] the same without contraction (hosts without AVX2, or baseline-only builds)
     *   2 [hip] sepFilter2D in Q8 integers (before 4.2: taps cvRound(k * 256) = 18 34 49 55 49 34 18, sum 257),
     *           (sum + 2^15) >> 16, saturated
     *   3 [hip] GaussianBlur's bit-exact fixed-point path (what a non-submatrix source takes): error-diffused
     *           taps 18 34 48 56 48 34 18 (sum 256), (sum + 2^15) >> 16                                */
    int32_t blur;
    /* resize(prev level, INTER_LINEAR_EXACT) inside ORB [OCV A.2]: rounding of the 8.8 coefficient
     *   0 [hip] cvRound(frac * 256), ties to even (softdouble -> ufixedpoint16)
     *   1 [hip] floor(frac * 256 + 0.5), ties up                                                      */
    int32_t resize;
    /* fastAtan2 in ORB's ICAngles [OCV A.5]: the odd degree-7 polynomial
     *   0 [hip] plain f32 multiplies and adds (scalar baseline code, SSE3: no fma)
     *   1 [hip] the Horner steps contracted to fma (a build whose BASELINE has FMA3)                  */
    int32_t atan;
    /* warpAffine(nearest, WARP_INVERSE_MAP), mo/lib.rs:339-347 [OCV A.10]
     *   0 [hip] 10-bit fixed point: (saturate<int>(M0 x * 1024) + saturate<int>((M1 y + M2) * 1024) + 512) >> 10
     *   1       cvRound of the f64 coordinate M0 x + M1 y + M2 (no fixed point; definitional cross-check) */
    int32_t warp;
    /* resize(INTER_AREA) tap construction, mo/image_utils.rs:17 [OCV A.11]
     *   0 [hip] computeResizeAreaTab: f32 weights, edge taps dropped below 1e-3 of a source cell
     *   1 [hip] exact box-overlap weights, no cut-off (definitional cross-check)                      */
    int32_t area;
    /* Levenberg-Marquardt step of estimateAffinePartial2D's refinement, mo/image_utils.rs:52 [OCV A.9]
     *   0 [hip] damped normal equations by Gaussian elimination with partial pivoting
     *   1       by Jacobi eigen-decomposition + back-substitution (cv::solve(DECOMP_EIG)); equal to f64 round-off */
    int32_t lm;
    /* cv::RNG multiplier (RANSAC sample schedule [OCV A.9], BRIEF pattern [OCV A.7]); any value [hip] */
    uint32_t rng_mul;             /* 4164903690 */
    /* verify_model 1 only: the homography of a point set, HomographyEstimatorCallback::runKernel of
     * calib3d/src/fundam.cpp (recalled; no counterpart in the reference, which never fits a homography)
     * DEFAULT 1 since ABI 6: the three forms give identical verdicts, survivals and inlier counts on all 9000 candidates of
     * the headline shape (profiles/r04_hdlt_agreement.json; form 2: 99.98 %), form 0 costs 7.5x the whole step, and there
     * is no reference behaviour to be faithful to — 0 stays as the switch for fidelity to cv::findHomography's rounding.
     *   0 [hip] normalised DLT: centroid / mean-absolute-deviation normalisation, the 9x9 normal matrix L^T L
     *           accumulated in f64, cv::eigen = the Jacobi sweep of core/src/lapack.cpp (JacobiImpl_, pivot =
     *           largest off-diagonal element, its own hypot), H = the eigenvector of the smallest eigenvalue,
     *           de-normalised and scaled by 1 / H[8]
     *   1 [hip] minimal samples (4 pairs) only: the 8x8 system with h33 = 1 on the same normalised points by
     *           Gaussian elimination with partial pivoting; point sets of more than 4 pairs (the refit over the
     *           inliers) still take form 0.  Agrees with form 0 to f64 round-off (same samples, same masks in every
     *           test) at 1 / 60 of its cost: a RANSAC iteration of form 0 is a 9x9 eigen-decomposition (~140
     *           Jacobi rotations).  The fast choice when fidelity to cv::eigen's rounding is not the point.
     *   2 [hip] minimal samples only: the same model in closed form — the projective maps of the unit square onto the two
     *           normalised quadrilaterals (Heckbert), H = S_to * adj(S_from): ~90 multiplications and two divisions, no
     *           pivoting; again equal to f64 round-off, and the cheapest of the three. */
    int32_t hdlt;
} slideo_ocv_variants;

/* Every literal the reference hard-codes on the hot path, as one struct whose
 * defaults (slideo_config_default) equal those literals. */
typedef struct slideo_config {
    /* cv::ORB::create arguments, mo/feature_extractor.rs:13-23 */
    int32_t nfeatures;            /* 2000 */
    float   scale_factor;         /* 1.2f */
    int32_t nlevels;              /* 8    */
    int32_t edge_threshold;       /* 62   */
    int32_t patch_size;           /* 62   */
    int32_t fast_threshold;       /* 20   */
    /* matcher: k of knn_match, mo/lib.rs:266 */
    int32_t knn_k;                /* 30   */
    /* tolerance vote, mo/lib.rs:275 */
    float   vote_tolerance;       /* 1.05f */
    /* candidate pages kept, mo/lib.rs:295 */
    int32_t max_candidate_pages;  /* 40   */
    /* estimate_affine_partial_2d arguments, mo/image_utils.rs:52 */
    double  ransac_threshold;     /* 3.0  */
    int32_t ransac_max_iters;     /* 2000 */
    double  ransac_confidence;    /* 0.99 */
    int32_t refine_iters;         /* 10   */
    /* rating filter, mo/lib.rs:330,333 */
    int32_t max_rated;            /* 10   */
    double  min_rating;           /* 50.0 (strict >) */
    double  min_rating_ratio;     /* 0.2  (strict >) */
    /* verdict, mo/lib.rs:381 */
    float   min_similarity;       /* 0.5f (strict >) */
    /* to_small_image, mo/image_utils.rs:11 */
    int32_t small_area;           /* 300*400 */
    /* MarkSimilarIter, mo/video_capture.rs:98 */
    float   changed_similarity;   /* 0.98f (changed <=> similarity < this) */
    /* Extension with no reference counterpart (BASELINE.json north_star / configs[1] wording): 0 = off, the
     * reference's tolerance vote above.  r > 0 replaces it by the ratio test on the two nearest neighbours:
     * a query votes for its nearest row iff it has a second neighbour and (float)d1 < r * (float)d2 (f32,
     * strict); needs knn_k >= 2. */
    float   ratio_test;           /* 0.0f */
    /* Extension with no reference counterpart (BASELINE.json north_star "RANSAC homography verification",
     * configs[4]): the geometric model of the verification step.
     *   0 = the reference's estimateAffinePartial2D (4-DOF similarity, 2-point samples, mo/image_utils.rs:45-60)
     *       followed by warpAffine (mo/lib.rs:339-347);
     *   1 = an 8-DOF homography: what cv::findHomography(from, to, RANSAC, ransac_threshold, mask,
     *       ransac_max_iters, ransac_confidence) computes (calib3d/src/fundam.cpp, recalled: 4-point samples drawn
     *       by RANSACPointSetRegistrator::getSubset with HomographyEstimatorCallback::checkSubset, normalised DLT,
     *       f32 re-projection error, the same cv::RNG(-1) schedule and sequential acceptance as the similarity
     *       path, then — refine_iters > 0 and more than 4 pairs — a DLT over all inliers and refine_iters
     *       Levenberg-Marquardt steps on the 8 parameters; the mask is not recomputed), followed by
     *       warpPerspective(nearest, WARP_INVERSE_MAP) (imgproc/src/imgwarp.cpp) in the re-projection. */
    int32_t verify_model;         /* 0 */
    /* The descriptor index.  0 = exact brute-force Hamming k-NN (north_star; what this library is built around).
     * 1 = LSH-compatible approximate search: the candidate rule of the index the reference really builds —
     * FlannBasedMatcher over FLANN's LshIndex with table_number 6, key_size 12, multi_probe_level 1 (mo/flann.rs:14-26):
     * lsh_tables hash tables, each keyed by lsh_key_bits descriptor bits (cv::randShuffle of the 256 bit positions on
     * cv::RNG's default state, the first lsh_key_bits of it, as flann/lsh_table.h does — recalled), a train row is a
     * CANDIDATE of a query iff in some table its key differs from the query's in at most lsh_multi_probe bits; the
     * result is the knn_k nearest candidates by (distance, row).  (FLANN's KNNUniqueResultSet breaks distance ties
     * at the k-th place by visiting order; here ties go to the lower row: canonical)  Recall < 1, like
     * the reference's; `SearchParams.checks` (32, mo/flann.rs:21) is ignored by LshIndex and has no counterpart. */
    int32_t matcher;              /* 0 */
    int32_t lsh_tables;           /* 6  (mo/flann.rs:16) */
    int32_t lsh_key_bits;         /* 12 (mo/flann.rs:17) */
    int32_t lsh_multi_probe;      /* 1  (mo/flann.rs:18) */
    /* Opt-in DEPARTURE from the reference's verdict (mo/lib.rs:370-389: survivors sorted by re-projection similarity, the
     * first above min_similarity wins).  0 = that rule.  1 = survivors keep their RATING order (inlier count descending, ties
     * in candidate order: mo/lib.rs:329) and the first whose similarity exceeds min_similarity wins — the similarity becomes
     * an acceptance test instead of the ranking.  Why it exists: with verify_model 1 a template-sharing sibling page
     * re-projects its shared template as well as the true page and wins by a few thousandths of similarity on ~13 % of the
     * synthetic headline frames, while the true page has the most inliers (DESIGN.md section 5 has the measured effect). */
    int32_t verdict_rule;         /* 0 */
    /* which restatement of each OpenCV primitive to run.  Defaults: all 0 EXCEPT hdlt = 1 (ABI 6), rng_mul 4164903690 — obtain
     * them from slideo_config_default: a zero-initialised ocv selects hdlt 0, a different (60x slower) form than the default */
    slideo_ocv_variants ocv;
} slideo_config;

/* cv::KeyPoint as the reference consumes it (pt, size, angle, response,
 * octave); class_id is never read (mo/lib.rs:299-302). 24 bytes. */
typedef struct slideo_keypoint {
    float   x, y;       /* level-0 pixel coordinates                         */
    float   size;       /* patch_size * scale(octave)                        */
    float   angle;      /* degrees, [0,360)                                  */
    float   response;   /* FAST score                                        */
    int32_t octave;     /* pyramid level                                     */
} slideo_keypoint;

/* One per-frame verdict = the `image: Option<I>` of matching::Matching
 * (crates/matching/src/lib.rs:35-40) plus the two numbers that decided it. */
typedef struct slideo_verdict {
    int32_t page_idx;     /* -1 = None                                        */
    float   similarity;   /* of the winning page, 0 when page_idx == -1       */
    int32_t inliers;      /* RANSAC inlier count of the winning page, else 0  */
    int32_t n_keypoints;  /* ORB keypoints found in the frame                 */
} slideo_verdict;

typedef struct slideo_matcher slideo_matcher;

/* matching::ProgressReporter (crates/matching/src/progress.rs:3-17).  May be
 * invoked from any thread; the reference invokes it from rayon workers
 * (mo/lib.rs:49-53,192-203). */
typedef void (*slideo_progress_fn)(void* user, uint64_t done, uint64_t total, const char* msg);

uint32_t    slideo_abi_version(void);

/* Fills *cfg with the reference's literals (table above). */
void        slideo_config_default(slideo_config* cfg);

/* Replaces OpenCVImageVideoMatcher::default() + FeatureExtractor::default()
 * (crates/app/src/main.rs:69; mo/feature_extractor.rs:12-27).
 * device: HIP device ordinal. */
int32_t     slideo_matcher_create(const slideo_config* cfg, int32_t device, slideo_matcher** out);
void        slideo_matcher_destroy(slideo_matcher* m);

/* Message of the last failure on `m` (or of the last failed create when m is
 * NULL).  Never NULL; valid until the next call on the same handle. */
const char* slideo_last_error(const slideo_matcher* m);

/* Replaces ProcessedImage::compute over all pages (mo/lib.rs:45-56,92-131):
 * ORB detect+describe and to_small_image per page.  Host buffers.  May be
 * called repeatedly before finalize; pages keep their add order. */
int32_t     slideo_matcher_add_pages_bgr8(slideo_matcher* m, int32_t n_pages,
                                          const uint8_t* const* data,
                                          const int32_t* width, const int32_t* height,
                                          const int32_t* stride_bytes);

/* The page side of ProcessedImage (mo/lib.rs:77-83) computed elsewhere — by another rank that analysed a share of the deck
 * (the page-sharded build of SURVEY.md section 8e: ranks all-gather these records instead of each analysing every page) or by
 * an earlier run (the reference caches per-PDF state, crates/app/src/db.rs).  Appends ONE page: its size, its keypoints and
 * descriptors in canonical order (as slideo_matcher_get_page_features returns them) and its small image (as
 * slideo_matcher_get_page_small returns it; small_w x small_h x 3 bytes, must be the to_small_image size of the page).
 * A matcher built from imported pages behaves exactly like one that analysed the images itself. */
int32_t     slideo_matcher_add_page_features(slideo_matcher* m, int32_t width, int32_t height, int32_t n_keypoints,
                                             const slideo_keypoint* kp, const uint8_t* desc32,
                                             const uint8_t* small_bgr, int32_t small_w, int32_t small_h);
/* Copies page `page_idx`'s small image (to_small_image, mo/lib.rs:128) to host; *sw, *sh receive its size. */
int32_t     slideo_matcher_get_page_small(const slideo_matcher* m, int32_t page_idx, uint8_t* out, int64_t out_capacity,
                                          int32_t* sw, int32_t* sh);

/* Replaces FlannMatcher::new (mo/flann.rs:65-71; add :28-34, train :45-47):
 * freezes the page descriptor set into the device-resident train matrix. */
int32_t     slideo_matcher_finalize_pages(slideo_matcher* m);

int32_t     slideo_matcher_page_count(const slideo_matcher* m);
/* Total descriptors over all pages (M).  -1 before finalize. */
int64_t     slideo_matcher_descriptor_count(const slideo_matcher* m);
/* Distinct descriptors among them (<= M).  The matcher's k-NN stage searches these and then restores, exactly, what a search
 * over all M rows returns (csrc/knn.hip.h knn_expand_dups_kernel): FlannMatcher::knn_match returns every matching row
 * (mo/flann.rs:73-89) and the vote counts per row (mo/lib.rs:268-282), so equal rows of different pages all vote.  Decks
 * repeat templates: 21 % of the rows of the 500-page benchmark deck are duplicates.  -1 before finalize. */
int64_t     slideo_matcher_unique_descriptor_count(const slideo_matcher* m);
/* Copies page `page_idx`'s keypoints/descriptors (canonical order) to host. */
int32_t     slideo_matcher_get_page_features(const slideo_matcher* m, int32_t page_idx,
                                             slideo_keypoint* kp, uint8_t* desc32,
                                             int32_t capacity, int32_t* n_out);

/* Replaces OpenCVVideoMatcherTask::match_images_with_frame (mo/lib.rs:249-413)
 * for a batch of equally sized frames held in HOST memory; verdicts to host. */
int32_t     slideo_match_frames_bgr8(slideo_matcher* m, int32_t n_frames,
                                     const uint8_t* frames, int32_t width, int32_t height,
                                     int32_t stride_bytes, int64_t frame_stride_bytes,
                                     slideo_verdict* verdicts_out);

/* Same, frames already resident in HBM (device pointer); verdicts to host.
 * `hip_stream` is a hipStream_t (NULL = the matcher's own stream). */
int32_t     slideo_match_frames_bgr8_dev(slideo_matcher* m, int32_t n_frames,
                                         const uint8_t* frames_dev, int32_t width, int32_t height,
                                         int32_t stride_bytes, int64_t frame_stride_bytes,
                                         slideo_verdict* verdicts_out, void* hip_stream);

/* Streaming form of the same call for callers that keep the GPU fed: submit a unit of frames (device
 * memory, must stay valid until collected), later collect its verdicts.  At most
 * slideo_matcher_max_in_flight() units (4) are in flight; each runs on its own internal stream and workspace
 * slot so that the ORB and verification stages of some units share the GPU with the matrix-core bound kNN of
 * others (the reference gets the same effect from rayon's work stealing, mo/lib.rs:174,213).  Tickets must
 * be collected in submission order.  The synchronous entry points above use the same machinery on two
 * halves of their batch. */
int32_t     slideo_matcher_max_in_flight(const slideo_matcher* m);
int32_t     slideo_match_frames_submit_dev(slideo_matcher* m, int32_t n_frames,
                                           const uint8_t* frames_dev, int32_t width, int32_t height,
                                           int32_t stride_bytes, int64_t frame_stride_bytes,
                                           void* hip_stream, int64_t* ticket_out);
int32_t     slideo_match_frames_collect(slideo_matcher* m, int64_t ticket, slideo_verdict* verdicts_out);
/* The same, and the unit's verdict records are also left in caller-provided DEVICE memory (n_frames records of
 * slideo_verdict, complete when the call returns): what a multi-GPU caller hands to its one all-gather of verdicts
 *  without a host round trip.  verdicts_dev_out may be NULL. */
int32_t     slideo_match_frames_collect_dev(slideo_matcher* m, int64_t ticket, slideo_verdict* verdicts_out, void* verdicts_dev_out);

/* Replaces MarkSimilarIter (mo/video_capture.rs:86-98) for a run of sampled
 * frames in host memory: changed[i] = 1 iff similarity(small(frame i-1),
 * small(frame i)) < cfg.changed_similarity; the first frame compares against
 * `prev_small` (w*h*3 small image returned by an earlier call) or, when that
 * is NULL, is always changed.  similarity_out may be NULL. */
int32_t     slideo_changed_mask_bgr8(slideo_matcher* m, int32_t n_frames,
                                     const uint8_t* frames, int32_t width, int32_t height,
                                     int32_t stride_bytes, int64_t frame_stride_bytes,
                                     const uint8_t* prev_small, uint8_t* last_small_out,
                                     uint8_t* changed_out, float* similarity_out);

/* The frames of the LAST slideo_changed_mask_bgr8 call are still on the device when it returns.  This matches a subset of
 * them — sel[i] = index into that call's frames, ascending or not — exactly as slideo_match_frames_bgr8 would match the same
 * frames, without uploading them a second time: what OpenCVVideoMatcherTask::process does with the frames MarkSimilarIter
 * flagged as changed (mo/lib.rs:205-214).  Must directly follow the mask call on this handle (any other call that uploads
 * frames invalidates the kept ones: SLIDEO_ERR_STATE). */
int32_t     slideo_match_kept_frames(slideo_matcher* m, int32_t n_sel, const int32_t* sel, slideo_verdict* verdicts_out);

/* Optional: page-lock a frame buffer the caller reuses across calls (hipHostRegister / hipHostUnregister), so that the H2D
 * copies read it by DMA directly.  Not needed for correctness; pageable memory is staged by the runtime. */
int32_t     slideo_host_register(void* ptr, size_t bytes);
int32_t     slideo_host_unregister(void* ptr);

/* Optional progress sink for add_pages / match_frames. */
int32_t     slideo_matcher_set_progress(slideo_matcher* m, slideo_progress_fn fn, void* user);


/* ---- measurement ---------------------------------------------------------- */

/* Stage timing with HIP events recorded on the stream the kernels are launched
 * on (bench.py's roofline figures come from here).  enable != 0 starts
 * accumulating; reading returns and clears the accumulators.  Stages:
 *   0 orb (gray..describe)  1 knn (knn_hamming_kernel [+ merge])  2 verify
 *   (vote, ransac, rate, reproject, verdict)  3 whole sub-batch incl. copies.
 * launches_out[i] = number of timed intervals of stage i; for stage 1 each
 * interval is exactly one knn_hamming_kernel launch (+ its merge when split). */
/* kNN engine: 0 = FP4 matrix cores, wave shape chosen per launch (default), 1 = integer-VALU popcount kernel,
 * 2 = FP4 matrix cores forced to 2 waves per SIMD x 4 query tiles per wave (leaves half of every SIMD's registers
 * and 96 KB of LDS per CU to the kernels of the other unit in flight; what 0 picks when the queries fill the chip),
 * 3 = forced to 4 waves per SIMD x 2 query tiles per wave (what 0 picks for smaller query sets).
 * All are exact and return identical results; the switch exists for A/B measurement. */
int32_t     slideo_matcher_set_knn_engine(slideo_matcher* m, int32_t engine);

/* The matcher's kNN stage is fused with the acceptance rule of its only consumer, the tolerance vote
 * (mo/lib.rs:268-282: a neighbour counts iff d < best * 1.05): by default it keeps the k-NN lists exact only
 * for the neighbours that can still pass that test (every such neighbour, in canonical order), which spares
 * most of the list maintenance.  Verdicts, votes and candidates are identical either way; on != 0 makes the
 * stage keep the full exact k-NN lists (A/B measurement, tests).  slideo_knn_hamming is always exact. */
int32_t     slideo_matcher_set_knn_exact_lists(slideo_matcher* m, int32_t on);

#define SLIDEO_N_STAGES 4
int32_t     slideo_matcher_set_profiling(slideo_matcher* m, int32_t enable);
int32_t     slideo_matcher_read_profile(slideo_matcher* m, double* ms_out /*[4]*/,
                                        int64_t* launches_out /*[4]*/, int64_t* knn_pairs_out);
/* ABI 7.  The shader clock (MHz) the search kernel's waves ran at since profiling was switched on or this was last read:
 * wave 0 of every 8th search block sums its s_memtime (shader cycles) and s_memrealtime (100 MHz) deltas in device memory
 * (csrc/knn_tile.hip.h KtClock); *samples_out = the blocks that recorded (0: no search ran while profiling; *mhz_out is 0
 * then).  Measurement only (bench.py roofline.shader_clock_mhz): the chip clocks to its power budget, 2.1 - 2.3 of 2.4 GHz
 * under this load.  The matcher must be idle. */
int32_t     slideo_matcher_read_shader_clock(slideo_matcher* m, double* mhz_out, int64_t* samples_out);

/* ---- debug taps used by the parity tests ------------------------------ */

/* FeatureExtractor::find_keypoints_and_descriptors (mo/feature_extractor.rs:29-46)
 * on one host image.  Output in canonical order (octave, y, x).  *n_out is the
 * number found even when it exceeds `capacity` (then SLIDEO_ERR_CAPACITY). */
int32_t     slideo_orb_bgr8(slideo_matcher* m, const uint8_t* bgr, int32_t width, int32_t height,
                            int32_t stride_bytes, slideo_keypoint* kp, uint8_t* desc32,
                            int32_t capacity, int32_t* n_out);

/* Pyramid level `level` (gray, unblurred or blurred) of one host image; out is
 * lw*lh bytes, dimensions returned in *lw,*lh. */
int32_t     slideo_pyramid_level_bgr8(slideo_matcher* m, const uint8_t* bgr, int32_t width,
                                      int32_t height, int32_t stride_bytes, int32_t level,
                                      int32_t blurred, uint8_t* out, int64_t out_capacity,
                                      int32_t* lw, int32_t* lh);

/* Exact Hamming k-NN (replaces FlannMatcher::knn_match, mo/flann.rs:73-89 with
 * the brute-force search north_star asks for).  q: nq*32 bytes, t: nt*32 bytes
 * (host).  Out: idx[nq*k] (global train row, -1 padding when nt<k) and
 * dist[nq*k] (65535 padding), ascending by (distance, train row). */
int32_t     slideo_knn_hamming(slideo_matcher* m, const uint8_t* q, int32_t nq,
                               const uint8_t* t, int32_t nt, int32_t k,
                               int32_t* idx_out, uint16_t* dist_out);

/* The LSH-compatible search (slideo_config.matcher 1) as a tap: the k nearest (distance, row) among the rows of t that are
 * LSH candidates of each query under cfg's lsh_* parameters.  Same output format as slideo_knn_hamming. */
int32_t     slideo_knn_lsh(slideo_matcher* m, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, int32_t k,
                           int32_t* idx_out, uint16_t* dist_out);

/* North-star extension without a counterpart in the reference (BASELINE configs[2], SURVEY §8(d) "cfg2" / §8(f) N4):
 * exact squared-L2 k-NN between 128-dimensional u8 descriptors (SIFT-shaped: OpenCV's SIFT descriptors are
 * integer-valued 0..255), computed as an N x M x 128 integer contraction on the matrix cores
 * (v_mfma_i32_32x32x32_i8 on centred components).  q: nq x 128 bytes, t: nt x 128 bytes (nt < 2^23), 1 <= k <= 32.
 * idx_out[nq*k]: train row or -1; dist_out[nq*k]: squared distance (0xFFFFFFFF where idx is -1).  Neighbours in
 * ascending (distance, row) order — what cv::BFMatcher(NORM_L2).knnMatch would return up to the square root.
 * The parity target is this repository's CPU restatement (oracle so_knn_l2_u8). */
int32_t     slideo_knn_l2_u8(slideo_matcher* m, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, int32_t k,
                             int32_t* idx_out, uint32_t* dist_out);

/* The same search with the train set prepared once and kept on the device (what a SIFT matcher over a page DB would hold:
 * the counterpart of FlannMatcher::new for float descriptors) and queries / results in device memory.  kernel_ms (may be
 * null) receives the HIP-event time of the search kernels of this call.  bench.py --workload cfg2 times this. */
int32_t     slideo_l2_set_train(slideo_matcher* m, const uint8_t* t, int32_t nt);
int32_t     slideo_l2_knn_dev(slideo_matcher* m, const void* q_dev, int32_t nq, int32_t k, void* idx_dev /* i32 [nq*k] */,
                              void* dist_dev /* u32 [nq*k] */, float* kernel_ms);

/* ---- SIFT (north-star extension, BASELINE configs[2]; no counterpart in the reference, whose only extractor is ORB:
 * mo/feature_extractor.rs:3-4,13).  cv::SIFT::detectAndCompute of OpenCV 4.5.2 as restated in oracle/sift_oracle.h (the
 * parity target; its header lists the two documented departures): doubled first octave, nOctaveLayers + 3 Gaussian layers per
 * octave, DoG extrema with sub-pixel refinement, contrast and edge tests, orientation histogram (one keypoint per peak),
 * retainBest(nfeatures) by response with ties kept, 4 x 4 x 8 descriptors as 128 bytes (OpenCV's are integer-valued 0..255).
 * Keypoints come back in canonical order (octave, layer, row, column, orientation bin); `octave` holds OpenCV's packed field
 * (octave & 255 | layer << 8 | sub-layer << 16), x / y / size in input-image pixels.  The descriptors feed slideo_l2_knn_dev.
 * Limits: n_octave_layers must be 3; image sides <= 4095 (the doubled image's coordinates travel in 13 bits). */
typedef struct slideo_sift_config {
    int32_t nfeatures;            /* 0 = keep every keypoint (cv::SIFT::create default)  */
    int32_t n_octave_layers;      /* 3    */
    double  contrast_threshold;   /* 0.04 */
    double  edge_threshold;       /* 10   */
    double  sigma;                /* 1.6  */
} slideo_sift_config;
void        slideo_sift_config_default(slideo_sift_config* cfg);
/* One host image.  *n_out = keypoints found even when it exceeds `capacity` (then SLIDEO_ERR_CAPACITY). */
int32_t     slideo_sift_bgr8(slideo_matcher* m, const slideo_sift_config* cfg, const uint8_t* bgr, int32_t width, int32_t height,
                             int32_t stride_bytes, slideo_keypoint* kp, uint8_t* desc128, int32_t capacity, int32_t* n_out);
/* A batch of equally sized frames in DEVICE memory -> keypoints and descriptors in DEVICE memory, frame after frame:
 * frame f owns rows [qofs_out[f], qofs_out[f + 1]) of kp_dev (slideo_keypoint) and desc_dev (128 bytes each).  qofs_out: n + 1
 * host values.  capacity_total rows must fit (else SLIDEO_ERR_CAPACITY).  kernel_ms (may be null): HIP-event time of the call's
 * kernels.  bench.py --workload cfg2 times this + slideo_l2_knn_dev. */
int32_t     slideo_sift_frames_dev(slideo_matcher* m, const slideo_sift_config* cfg, int32_t n_frames, const uint8_t* frames_dev,
                                   int32_t width, int32_t height, int32_t stride_bytes, int64_t frame_stride_bytes,
                                   int64_t capacity_total, void* kp_dev, void* desc_dev, uint32_t* qofs_out, float* kernel_ms);
/* Pyramid tap (parity tests): Gaussian layer (dog == 0, layer 0 .. 5) or difference layer (dog != 0, layer 0 .. 4) of `octave`. */
int32_t     slideo_sift_layer_bgr8(slideo_matcher* m, const slideo_sift_config* cfg, const uint8_t* bgr, int32_t width, int32_t height,
                                   int32_t stride_bytes, int32_t octave, int32_t layer, int32_t dog, float* out, int64_t out_capacity,
                                   int32_t* lw, int32_t* lh);

/* North-star / BASELINE configs[2] as a COMPLETE matcher (no reference counterpart: the reference extracts ORB only,
 * mo/feature_extractor.rs:3-4,13): SIFT-128 features instead of ORB for pages and frames, brute-force squared-L2 k-NN (k = 2) on
 * the int8 matrix cores with Lowe's ratio test — a query votes for its nearest row iff sqrt(d1) < ratio * sqrt(d2) (f32) —
 * instead of the Hamming k-NN + tolerance vote.  Everything from the per-page vote on (candidate ranking, RANSAC per
 * verify_model, rating, re-projection, verdict) and every entry point (add_pages, finalize, match_frames*, submit / collect,
 * changed mask + kept frames, the candidate trace) is the path's own.  Must be called before the first page is added; cfg as
 * for slideo_sift_bgr8 (nfeatures 0 = all keypoints), 0 < ratio <= 1.  ratio == 0: no ratio test — the path's own tolerance
 * vote (mo/lib.rs:268-282) on the knn_k nearest rows instead: a neighbour counts iff sqrt(d) < sqrt(d_best) * vote_tolerance
 * (f32), which — unlike Lowe's test — keeps the matches whose descriptor also sits on a twin page of the same template.
 * slideo_matcher_get_page_features (here 128 bytes per keypoint) works; slideo_matcher_
 * add_page_features (32-byte descriptors) returns SLIDEO_ERR_UNSUPPORTED in this mode; descriptor_count is the number of SIFT
 * rows.  Parity target: the oracle's so_db_use_sift + so_match_frame. */
int32_t     slideo_matcher_use_sift(slideo_matcher* m, const slideo_sift_config* cfg, float ratio);

/* to_small_image (mo/image_utils.rs:8-20) of one host image. */
int32_t     slideo_small_image_bgr8(slideo_matcher* m, const uint8_t* bgr, int32_t width,
                                    int32_t height, int32_t stride_bytes, uint8_t* out,
                                    int64_t out_capacity, int32_t* sw, int32_t* sh);

/* Per-frame trace of the decision steps for one frame of the LAST
 * slideo_match_frames_* call (parity tests compare these with the oracle). */
typedef struct slideo_candidate {
    int32_t page_idx;
    int32_t n_votes;      /* matches surviving the tolerance vote (mo/lib.rs:268-282) */
    int32_t inliers;      /* rating (mo/lib.rs:310)                                    */
    int32_t survived;     /* passed the rating filter (mo/lib.rs:333)                  */
    float   similarity;   /* mo/lib.rs:351, 0 when not computed                        */
    double  transform[9]; /* 3x3 row-major, slide -> frame; verify_model 0: rows 0-1 = the 2x3 of
                             estimateAffinePartial2D (mo/image_utils.rs:52), row 2 = 0 0 1 (all 0 when no model was found) */
} slideo_candidate;

int32_t     slideo_last_frame_candidates(const slideo_matcher* m, int32_t frame_in_batch,
                                         slideo_candidate* out, int32_t capacity, int32_t* n_out);

/* ---- N-device group ----------------------------------------------------------------------------------------------
 * One matcher per device behind ONE handle: what the reference's fan-out over every core of the machine becomes on a node
 * with several GPUs (rayon: one task per changed frame, mo/lib.rs:174,213; pages par_iter, mo/lib.rs:45-47).  The page
 * database is replicated on every member device, a call's frames are cut into contiguous shards — member r
 * takes the r-th contiguous block, the blocks differing in size by at most one (the first n mod N take one more) — each shard runs through its device's matcher
 * on a host thread of its own, and every shard's verdicts land in the caller's array at the shard's offset: the gather of the
 * in-process form is the device-to-host copy each member makes anyway.  Page analysis is sharded the same way and every
 * member appends the whole call in page order.  Results are those of a single matcher, bit for bit, whatever N is.
 * `devices`: HIP ordinals, one member each (an ordinal may repeat: two members then share a device).  A group is not
 * re-entrant (like a matcher); progress callbacks fire from the member threads, one at a time, with a count that never
 * decreases.  (One PROCESS per GPU — where every rank needs
 * the whole timeline — is the other multi-GPU form: slideo_match_frames_collect_dev leaves a rank's records in device memory
 * for ONE RCCL all-gather, bench.py / slideo_amd/distributed.py.) */
typedef struct slideo_group slideo_group;
/* gfx950 devices visible to this process (0 when there is none: nothing here runs without one). */
int32_t     slideo_device_count(void);
/* Their HIP ordinals, ascending: fills at most `capacity` entries of ordinals_out (may be NULL) and returns how many there
 * are.  On a node whose HIP ordinals also name other architectures the ordinals are not 0 .. count-1. */
int32_t     slideo_device_list(int32_t* ordinals_out, int32_t capacity);
/* n_devices == 0 (devices may then be NULL): one member per gfx950 device of the node = slideo_device_list's ordinals. */
int32_t     slideo_group_create(const slideo_config* cfg, int32_t n_devices, const int32_t* devices, slideo_group** out);
void        slideo_group_destroy(slideo_group* g);
/* Message of the last failure on `g` (of the last failed create when g is NULL); names the member and its device. */
const char* slideo_group_last_error(const slideo_group* g);
int32_t     slideo_group_device_count(const slideo_group* g);
/* Member i's matcher (owned by the group): for the introspection calls, the taps and the measurement hooks above. */
slideo_matcher* slideo_group_member(slideo_group* g, int32_t i);
int32_t     slideo_group_set_progress(slideo_group* g, slideo_progress_fn fn, void* user);
/* slideo_matcher_use_sift on every member (before the first page). */
int32_t     slideo_group_use_sift(slideo_group* g, const slideo_sift_config* cfg, float ratio);
/* slideo_matcher_add_pages_bgr8 with the call's pages analysed across the members (mo/lib.rs:45-56). */
int32_t     slideo_group_add_pages_bgr8(slideo_group* g, int32_t n_pages, const uint8_t* const* data,
                                        const int32_t* width, const int32_t* height, const int32_t* stride_bytes);
int32_t     slideo_group_finalize_pages(slideo_group* g);
int32_t     slideo_group_page_count(const slideo_group* g);
int64_t     slideo_group_descriptor_count(const slideo_group* g);
/* slideo_match_frames_bgr8 with the frames sharded over the members (mo/lib.rs:213-214: one independent task per frame). */
int32_t     slideo_group_match_frames_bgr8(slideo_group* g, int32_t n_frames, const uint8_t* frames, int32_t width, int32_t height,
                                           int32_t stride_bytes, int64_t frame_stride_bytes, slideo_verdict* verdicts_out);
/* Trace of frame `frame_in_batch` of the LAST slideo_group_match_frames_bgr8 call (from the member that matched it). */
int32_t     slideo_group_last_frame_candidates(const slideo_group* g, int32_t frame_in_batch, slideo_candidate* out, int32_t capacity,
                                               int32_t* n_out);
/* slideo_changed_mask_bgr8 over the members: a shard reads the one frame before its block (the previous sampled frame
 * MarkSimilarIter compares with, mo/video_capture.rs:86-98) — flags equal the single matcher's.  Every member keeps its
 * block's frames for slideo_group_match_kept_frames, which matches each selected frame on the member that holds it. */
int32_t     slideo_group_changed_mask_bgr8(slideo_group* g, int32_t n_frames, const uint8_t* frames, int32_t width, int32_t height,
                                           int32_t stride_bytes, int64_t frame_stride_bytes, const uint8_t* prev_small,
                                           uint8_t* last_small_out, uint8_t* changed_out, float* similarity_out);
int32_t     slideo_group_match_kept_frames(slideo_group* g, int32_t n_sel, const int32_t* sel, slideo_verdict* verdicts_out);

#ifdef __cplusplus
}
#endif
#endif /* SLIDEO_AMD_H */
