//! matching-hip — hediet/slideo's `matching` trait surface (crates/matching/src/lib.rs:7-40) over the MI355X-native
//! matcher libslideo_amd.so.  A drop-in for crates/matching-opencv: same traits, same progress protocol, same
//! panics-instead-of-Results, same post-processing of the per-frame results.
//!
//!   OpenCVImageVideoMatcher::default()          -> HipImageVideoMatcher::default()
//!   create_video_matcher    (mo/lib.rs:37-64)   -> slideo_group_create (one matcher per GPU of the node) + add_pages_bgr8 +
//!                                                  finalize_pages: pages analysed across the devices, DB replicated
//!   match_images_with_video (mo/lib.rs:140-158) -> opens the video to size the progress bar, as the reference does
//!   process                 (mo/lib.rs:168-246) -> sampled frames in batches, each batch sharded over the devices:
//!                                                  slideo_group_changed_mask_bgr8 (MarkSimilarIter,
//!                                                  mo/video_capture.rs:86-98), then slideo_group_match_kept_frames on
//!                                                  the changed ones (the mask's upload, on the device that holds it)
//!                                                  (match_images_with_frame, mo/lib.rs:249-413); sentinel, sort,
//!                                                  consecutive-duplicate removal as mo/lib.rs:185-189,229-244
//! (mo/ = crates/matching-opencv/src/)
//!
//! UNCOMPILED in the repository this crate ships in (no Rust toolchain there).
mod decode;
mod ffi;

use matching::{
    ImageVideoMatcher, MatchableImage, Matching, ProgressReporter, VideoMatcher, VideoMatcherTask,
};
use std::{
    ffi::CStr,
    path::{Path, PathBuf},
    sync::{Arc, Mutex},
    time::Duration,
};

/// Sampled frames handed to the library per call AND DEVICE.  The changed-frame mask needs consecutive samples in one call
/// (or the previous small image carried over, which is what happens at batch seams); 64 x 1080p = 400 MB of host memory
/// per device of the group.
const FRAMES_PER_CALL_PER_DEVICE: usize = 64;

struct RawHandle(*mut ffi::slideo_group);
// The pointer itself may move between threads; USE is serialised by the Mutex below — include/slideo_amd.h: "a matcher
// is NOT re-entrant: calls on one handle must come from one thread at a time".
unsafe impl Send for RawHandle {}

struct Handle {
    raw: Mutex<RawHandle>,
}

impl Drop for Handle {
    fn drop(&mut self) {
        let g = self.raw.lock().unwrap_or_else(|e| e.into_inner());
        unsafe { ffi::slideo_group_destroy(g.0) }
    }
}

/// The reference has no `Result` anywhere on this surface: it panics (mo/lib.rs:95-104, unwrap() throughout).
fn check(h: *mut ffi::slideo_group, rc: i32) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(ffi::slideo_group_last_error(h)) }
            .to_string_lossy()
            .into_owned();
        panic!("slideo_amd error {}: {}", rc, msg);
    }
}

pub struct HipImageVideoMatcher {
    /// HIP device ordinals, one matcher each (slideo_group_create).  Empty (default) = every gfx950 device of the node:
    /// the reference fans out over the whole machine too (the global rayon pool, mo/lib.rs:45,174).
    pub devices: Vec<i32>,
    /// Some(ratio): the north-star's SIFT + L2 front end (slideo_group_use_sift) instead of the reference's ORB + Hamming;
    /// Some(0.0) keeps the path's own 5 % tolerance vote on the L2 distances (the robust choice on decks whose pages
    /// share a template), Some(r > 0) is Lowe's ratio test.  None (default) = the reference's extractor and matcher.
    pub sift_ratio: Option<f32>,
}

impl Default for HipImageVideoMatcher {
    fn default() -> Self {
        HipImageVideoMatcher { devices: Vec::new(), sift_ratio: None }
    }
}

impl<'i> ImageVideoMatcher<'i> for HipImageVideoMatcher {
    fn create_video_matcher<I: MatchableImage + Send + Sync + Copy + Eq + 'i>(
        &self,
        images: Vec<I>,
        progress_reporter: ProgressReporter,
    ) -> Box<dyn VideoMatcher<'i, I> + 'i> {
        ffi::assert_abi();
        let len = images.len() as u64;
        let mut cfg = std::mem::MaybeUninit::<ffi::slideo_config>::uninit();
        let mut h: *mut ffi::slideo_group = std::ptr::null_mut();
        // no explicit list: n_devices 0 = every gfx950 device of the node, enumerated by the library by its own HIP
        // ordinals (no device at all: create reports it)
        let devices_ptr = if self.devices.is_empty() { std::ptr::null() } else { self.devices.as_ptr() };
        unsafe {
            ffi::slideo_config_default(cfg.as_mut_ptr()); // the reference's literals (mo/feature_extractor.rs:13-23 etc.)
            check(
                std::ptr::null_mut(),
                ffi::slideo_group_create(cfg.as_ptr(), self.devices.len() as i32, devices_ptr, &mut h),
            );
            if let Some(ratio) = self.sift_ratio {
                let mut sc = std::mem::MaybeUninit::<ffi::slideo_sift_config>::uninit();
                ffi::slideo_sift_config_default(sc.as_mut_ptr());
                check(h, ffi::slideo_group_use_sift(h, sc.as_ptr(), ratio));
            }
        }
        // Page analysis (mo/lib.rs:43-58).  Pages are decoded on the host and handed over in groups, so that a 1000-page
        // deck does not sit decoded in memory at once; the progress protocol is the reference's and is driven from here
        // (slideo_group_set_progress — a C callback for callers without a reporter of their own — is not needed).
        progress_reporter.report(0, len, "Analyzing PDF pages...");
        let mut done = 0u64;
        let n_members = unsafe { ffi::slideo_group_device_count(h) }.max(1) as usize;
        for group in images.chunks(32 * n_members) {
            let decoded: Vec<decode::BgrImage> =
                group.iter().map(|i| decode::decode_page_bgr(i.get_path())).collect();
            let ptrs: Vec<*const u8> = decoded.iter().map(|d| d.data.as_ptr()).collect();
            let w: Vec<i32> = decoded.iter().map(|d| d.width).collect();
            let hh: Vec<i32> = decoded.iter().map(|d| d.height).collect();
            let stride: Vec<i32> = w.iter().map(|w| w * 3).collect();
            unsafe {
                check(
                    h,
                    ffi::slideo_group_add_pages_bgr8(
                        h,
                        ptrs.len() as i32,
                        ptrs.as_ptr(),
                        w.as_ptr(),
                        hh.as_ptr(),
                        stride.as_ptr(),
                    ),
                );
            }
            // one report per page, counting up (mo/lib.rs:49-53; the reference's come from rayon workers in any order)
            for _ in group {
                done += 1;
                progress_reporter.report(done, len, "Analyzing PDF pages...");
            }
        }
        unsafe { check(h, ffi::slideo_group_finalize_pages(h)) }; // FlannMatcher::new, mo/flann.rs:65-71
        progress_reporter.report(len, len, "PDF page analysis successful."); // mo/lib.rs:58
        Box::new(HipVideoMatcher {
            handle: Arc::new(Handle { raw: Mutex::new(RawHandle(h)) }),
            images: Arc::new(images),
            n_devices: n_members,
        })
    }
}

struct HipVideoMatcher<I> {
    handle: Arc<Handle>,
    images: Arc<Vec<I>>,
    n_devices: usize,
}

impl<'i, I: MatchableImage + Send + Sync + Copy + Eq + 'i> VideoMatcher<'i, I> for HipVideoMatcher<I> {
    fn match_images_with_video(
        &self,
        video_path: &Path,
        progress_reporter: ProgressReporter,
    ) -> Box<dyn VideoMatcherTask<I> + 'i> {
        let interval = Duration::from_secs(5); // mo/lib.rs:145
        let vid = decode::sampled_video(video_path, interval);
        let total_time = vid.total_time();
        let frames_to_process = (total_time.as_secs_f64() / interval.as_secs_f64()) as u64; // mo/lib.rs:148
        progress_reporter.report(0, frames_to_process, ""); // mo/lib.rs:150
        Box::new(HipVideoMatcherTask {
            handle: self.handle.clone(),
            images: self.images.clone(),
            n_devices: self.n_devices,
            video_path: video_path.to_owned(),
            progress_reporter,
        })
    }
}

struct HipVideoMatcherTask<I> {
    handle: Arc<Handle>,
    images: Arc<Vec<I>>,
    n_devices: usize,
    video_path: PathBuf,
    progress_reporter: ProgressReporter,
}

impl<I: MatchableImage + Send + Sync + Copy + Eq> VideoMatcherTask<I> for HipVideoMatcherTask<I> {
    fn process(&self) -> Vec<Matching<I>> {
        let interval = Duration::from_secs(5); // mo/lib.rs:175
        let vid = decode::sampled_video(&self.video_path, interval);
        let total_time = vid.total_time();
        let total_frames = vid.total_frames();
        let frames_to_process = (total_time.as_secs_f64() / interval.as_secs_f64()) as u32; // mo/lib.rs:179
        let message = format!(
            "Processing frames of '{}'...",
            self.video_path.file_name().unwrap().to_string_lossy()
        );

        // "Add a matching to indicate the last frame." (mo/lib.rs:185-189)
        let mut results: Vec<Matching<I>> = vec![Matching {
            image: None,
            video_frame_idx: total_frames as usize,
            video_time: total_time,
        }];

        let guard = self.handle.raw.lock().unwrap(); // one caller at a time on the handle
        let h = guard.0;
        let mut progress = 0u64;
        let mut prev_small: Option<Vec<u8>> = None;
        // one call = one shard of FRAMES_PER_CALL_PER_DEVICE sampled frames per device (mo/lib.rs:213: one task per frame over the pool)
        for batch in decode::batches(vid, FRAMES_PER_CALL_PER_DEVICE * self.n_devices) {
            let n = batch.meta.len();
            let fb = batch.frame_bytes();
            // MarkSimilarIter (mo/video_capture.rs:86-98): changed <=> similarity to the previous SAMPLED frame < 0.98; the
            // first frame of the video is always changed (prev_small == None); the last small image of this call is the
            // `prev` of the next one.
            let mut changed = vec![0u8; n];
            let mut last_small = vec![0u8; small_image_bytes(batch.width, batch.height)];
            unsafe {
                check(
                    h,
                    ffi::slideo_group_changed_mask_bgr8(
                        h,
                        n as i32,
                        batch.frames.as_ptr(),
                        batch.width,
                        batch.height,
                        batch.width * 3,
                        fb as i64,
                        prev_small.as_ref().map_or(std::ptr::null(), |p| p.as_ptr()),
                        last_small.as_mut_ptr(),
                        changed.as_mut_ptr(),
                        std::ptr::null_mut(),
                    ),
                );
            }
            // a change of frame size (never within one video file) restarts the comparison, as a fresh Mat size would
            // make compute_similarity panic in the reference
            prev_small = Some(last_small);

            // the changed frames through match_images_with_frame (mo/lib.rs:213-214): the mask call left every device's block
            // of the upload on that device, slideo_group_match_kept_frames matches a subset by index — no second copy over PCIe
            let sel: Vec<i32> = (0..n as i32).filter(|&i| changed[i as usize] != 0).collect();
            let mut verdicts = vec![ffi::slideo_verdict::default(); sel.len()];
            if !sel.is_empty() {
                unsafe {
                    check(
                        h,
                        ffi::slideo_group_match_kept_frames(h, sel.len() as i32, sel.as_ptr(), verdicts.as_mut_ptr()),
                    );
                }
            }
            for (&i, v) in sel.iter().zip(verdicts.iter()) {
                let (time, frame_idx) = batch.meta[i as usize];
                results.push(Matching {
                    video_time: time,
                    video_frame_idx: frame_idx,
                    image: if v.page_idx >= 0 { Some(self.images[v.page_idx as usize]) } else { None },
                });
            }
            // one report per sampled frame, changed or not (mo/lib.rs:191-212)
            for _ in 0..n {
                progress += 1;
                self.progress_reporter.report(progress, frames_to_process as u64, &message);
            }
        }
        drop(guard);

        self.progress_reporter.report(
            frames_to_process as u64,
            frames_to_process as u64,
            &format!("Finished!"),
        ); // mo/lib.rs:223-227

        timeline(results)
    }
}

/// The task's result as the application stores it (crates/app/src/db.rs:162-191): ordered by time, and a page (or "no page")
/// listed only where it CHANGES — the reference's sort + removal of consecutive duplicates, mo/lib.rs:229-244.  `sort_by_key`
/// is stable, so the end-of-video sentinel and equal-time entries keep their push order, as there.
fn timeline<I: Clone + Eq>(mut results: Vec<Matching<I>>) -> Vec<Matching<I>> {
    results.sort_by_key(|m| m.video_time);
    results.dedup_by(|next, kept| next.image == kept.image); // (Vec::dedup_by keeps the first of a run, drops the followers)
    results
}

/// to_small_image's output size (mo/image_utils.rs:8-20) times 3 channels: what slideo_changed_mask_bgr8 writes to
/// `last_small_out`.
fn small_image_bytes(width: i32, height: i32) -> usize {
    let max_area = 300 * 400;
    let factor = ((max_area as f32) / ((width * height) as f32)).sqrt();
    let sw = ((width as f32) * factor) as i32;
    let sh = ((height as f32) * factor) as i32;
    (sw as usize) * (sh as usize) * 3
}
