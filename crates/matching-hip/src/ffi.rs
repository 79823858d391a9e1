//! Hand-written declarations of include/slideo_amd.h (ABI 7) — what bindgen would emit for the entry points this crate
//! uses.  Field order and types mirror the C structs exactly; tests/test_capi_load.py pins the C side's layout
//! (sizeof(slideo_config) == 168) and `assert_abi()` below pins the version at run time.
#![allow(non_camel_case_types)]
use std::os::raw::c_char;

pub const SLIDEO_ABI_VERSION: u32 = 7;

/// slideo_ocv_variants: which restatement of each OpenCV primitive runs.  slideo_config_default fills it; the
/// application never touches it.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct slideo_ocv_variants {
    pub gray: i32,
    pub blur: i32,
    pub resize: i32,
    pub atan: i32,
    pub warp: i32,
    pub area: i32,
    pub lm: i32,
    pub rng_mul: u32,
    pub hdlt: i32,
}

/// slideo_config: every literal the reference hard-codes on the hot path; defaults = those literals.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct slideo_config {
    pub nfeatures: i32,
    pub scale_factor: f32,
    pub nlevels: i32,
    pub edge_threshold: i32,
    pub patch_size: i32,
    pub fast_threshold: i32,
    pub knn_k: i32,
    pub vote_tolerance: f32,
    pub max_candidate_pages: i32,
    pub ransac_threshold: f64,
    pub ransac_max_iters: i32,
    pub ransac_confidence: f64,
    pub refine_iters: i32,
    pub max_rated: i32,
    pub min_rating: f64,
    pub min_rating_ratio: f64,
    pub min_similarity: f32,
    pub small_area: i32,
    pub changed_similarity: f32,
    /// 0.0 = the reference's tolerance vote (default); > 0: ratio test instead (extension)
    pub ratio_test: f32,
    /// 0 = the reference's estimateAffinePartial2D (default); 1 = 8-DOF homography (extension)
    pub verify_model: i32,
    /// 0 = exact brute-force k-NN (default); 1 = the LSH candidate rule of the reference's FLANN index
    pub matcher: i32,
    pub lsh_tables: i32,
    pub lsh_key_bits: i32,
    pub lsh_multi_probe: i32,
    /// 0 = the reference's verdict (best similarity, mo/lib.rs:370-389; default); 1 = rating order, similarity only accepts
    pub verdict_rule: i32,
    pub ocv: slideo_ocv_variants,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct slideo_verdict {
    /// -1 = None
    pub page_idx: i32,
    pub similarity: f32,
    pub inliers: i32,
    pub n_keypoints: i32,
}

/// slideo_sift_config: cv::SIFT::create's parameters (the north-star's optional extractor, slideo_matcher_use_sift)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct slideo_sift_config {
    pub nfeatures: i32,
    pub n_octave_layers: i32,
    pub contrast_threshold: f64,
    pub edge_threshold: f64,
    pub sigma: f64,
}

/// The N-device group (include/slideo_amd.h, "N-device group"): one matcher per GPU behind one handle.  This crate binds the
/// group form of every call; a group over one device is the single matcher.
#[repr(C)]
pub struct slideo_group {
    _private: [u8; 0],
}

extern "C" {
    pub fn slideo_abi_version() -> u32;
    pub fn slideo_config_default(cfg: *mut slideo_config);
    /// members of a group (n_devices 0 at create = every gfx950 device of the node)
    pub fn slideo_group_device_count(g: *const slideo_group) -> i32;
    pub fn slideo_group_create(
        cfg: *const slideo_config,
        n_devices: i32,
        devices: *const i32,
        out: *mut *mut slideo_group,
    ) -> i32;
    pub fn slideo_group_destroy(g: *mut slideo_group);
    pub fn slideo_group_last_error(g: *const slideo_group) -> *const c_char;
    pub fn slideo_group_add_pages_bgr8(
        g: *mut slideo_group,
        n_pages: i32,
        data: *const *const u8,
        width: *const i32,
        height: *const i32,
        stride_bytes: *const i32,
    ) -> i32;
    pub fn slideo_group_finalize_pages(g: *mut slideo_group) -> i32;
    pub fn slideo_group_match_frames_bgr8(
        g: *mut slideo_group,
        n_frames: i32,
        frames: *const u8,
        width: i32,
        height: i32,
        stride_bytes: i32,
        frame_stride_bytes: i64,
        verdicts_out: *mut slideo_verdict,
    ) -> i32;
    pub fn slideo_sift_config_default(cfg: *mut slideo_sift_config);
    /// optional: SIFT + L2 search in front of the verify stages (before the first page); ratio 0 = the path's tolerance vote
    pub fn slideo_group_use_sift(g: *mut slideo_group, cfg: *const slideo_sift_config, ratio: f32) -> i32;
    pub fn slideo_group_match_kept_frames(
        g: *mut slideo_group,
        n_sel: i32,
        sel: *const i32,
        verdicts_out: *mut slideo_verdict,
    ) -> i32;
    pub fn slideo_group_changed_mask_bgr8(
        g: *mut slideo_group,
        n_frames: i32,
        frames: *const u8,
        width: i32,
        height: i32,
        stride_bytes: i32,
        frame_stride_bytes: i64,
        prev_small: *const u8,
        last_small_out: *mut u8,
        changed_out: *mut u8,
        similarity_out: *mut f32,
    ) -> i32;
}

/// The struct layouts above are only valid for one ABI version of the library.
pub fn assert_abi() {
    let v = unsafe { slideo_abi_version() };
    assert_eq!(
        v, SLIDEO_ABI_VERSION,
        "libslideo_amd.so has ABI {} but this crate was written for ABI {}",
        v, SLIDEO_ABI_VERSION
    );
    assert_eq!(std::mem::size_of::<slideo_config>(), 168);
    assert_eq!(std::mem::size_of::<slideo_verdict>(), 16);
}
