//! Host-side decode helpers: page PNGs and sampled video frames, through the OpenCV modules the application already
//! links (imgcodecs, videoio).  No matching logic lives here.
//!
//! The sampling rule is the reference's, crates/matching-opencv/src/video_capture.rs:42-57: every frame is grabbed, a
//! frame is retrieved when `frame_idx % floor(fps * interval) < 1`, its time is `frame_idx / fps`.
use opencv::{
    core::Mat,
    imgcodecs::{imread, IMREAD_COLOR},
    prelude::*,
    videoio::{VideoCapture, CAP_PROP_FPS, CAP_PROP_FRAME_COUNT, CAP_PROP_POS_FRAMES},
};
use std::{path::Path, time::Duration};

/// One page as the library wants it: 8-bit BGR interleaved, tightly packed rows.
pub struct BgrImage {
    pub data: Vec<u8>,
    pub width: i32,
    pub height: i32,
}

/// crates/matching-opencv/src/lib.rs:92-104 panics when the file is missing or unreadable; so does this.
/// (The reference reads with flag 0 = grayscale and then asks for BGRA2BGR, which cannot both hold — SURVEY F10;
/// every later step needs 3-channel pages, so the page is read as 8UC3 BGR.)
pub fn decode_page_bgr(path: &Path) -> BgrImage {
    if !path.exists() {
        panic!("File '{:?}' must exist", path);
    }
    let mat = imread(&path.to_string_lossy(), IMREAD_COLOR).unwrap();
    if mat.empty().unwrap() {
        panic!("Could not read file '{:?}'", path);
    }
    mat_to_bgr(&mat)
}

/// Copies an 8UC3 Mat into a tightly packed buffer (a Mat's rows may be padded or it may be a view).
pub fn mat_to_bgr(mat: &Mat) -> BgrImage {
    assert_eq!(mat.channels().unwrap(), 3, "expected an 8UC3 image");
    let (w, h) = (mat.cols(), mat.rows());
    let row = (w as usize) * 3;
    let mut data = vec![0u8; row * h as usize];
    if mat.is_continuous().unwrap() {
        data.copy_from_slice(&mat.data_bytes().unwrap()[..row * h as usize]);
    } else {
        for y in 0..h {
            let src = mat.ptr(y).unwrap();
            let dst = &mut data[(y as usize) * row..(y as usize + 1) * row];
            unsafe { std::ptr::copy_nonoverlapping(src, dst.as_mut_ptr(), row) };
        }
    }
    BgrImage { data, width: w, height: h }
}

/// VideoCaptureIter of the reference (video_capture.rs:10-57), yielding packed BGR frames.
pub struct SampledVideo {
    video: VideoCapture,
    fps: f64,
    interval: Duration,
}

pub struct SampledFrame {
    pub image: BgrImage,
    pub time: Duration,
    pub frame_idx: usize,
}

impl SampledVideo {
    pub fn open(path: &Path, interval: Duration) -> Self {
        let video = VideoCapture::from_file(&path.to_string_lossy(), 0).unwrap();
        let fps = video.get(CAP_PROP_FPS).unwrap();
        SampledVideo { video, fps, interval }
    }

    pub fn total_frames(&self) -> f64 {
        self.video.get(CAP_PROP_FRAME_COUNT).unwrap()
    }

    pub fn total_time(&self) -> Duration {
        Duration::from_secs_f64(self.video.get(CAP_PROP_FRAME_COUNT).unwrap() / self.fps)
    }
}

impl Iterator for SampledVideo {
    type Item = SampledFrame;

    fn next(&mut self) -> Option<SampledFrame> {
        let mut frame = Mat::default();
        loop {
            let frame_idx = self.video.get(CAP_PROP_POS_FRAMES).unwrap();
            let time_passed = Duration::from_secs_f64(frame_idx / self.fps);
            if !self.video.grab().unwrap() {
                return None;
            }
            if frame_idx % (self.fps * self.interval.as_secs_f64()).floor() < 1.0 {
                self.video.retrieve(&mut frame, 0).unwrap();
                return Some(SampledFrame {
                    image: mat_to_bgr(&frame),
                    time: time_passed,
                    frame_idx: frame_idx as usize,
                });
            }
        }
    }
}

/// Up to `max` consecutive sampled frames of equal size, packed back to back (what slideo_changed_mask_bgr8 and
/// slideo_match_frames_bgr8 take).
pub struct FrameBatch {
    pub frames: Vec<u8>,
    pub meta: Vec<(Duration, usize)>,
    pub width: i32,
    pub height: i32,
}

impl FrameBatch {
    pub fn frame_bytes(&self) -> usize {
        self.width as usize * self.height as usize * 3
    }
}

pub struct Batches<I: Iterator<Item = SampledFrame>> {
    iter: std::iter::Peekable<I>,
    max: usize,
}

pub fn batches<I: Iterator<Item = SampledFrame>>(iter: I, max: usize) -> Batches<I> {
    Batches { iter: iter.peekable(), max: max.max(1) }
}

impl<I: Iterator<Item = SampledFrame>> Iterator for Batches<I> {
    type Item = FrameBatch;

    fn next(&mut self) -> Option<FrameBatch> {
        let first = self.iter.next()?;
        let (w, h) = (first.image.width, first.image.height);
        let mut b = FrameBatch { frames: first.image.data, meta: vec![(first.time, first.frame_idx)], width: w, height: h };
        while b.meta.len() < self.max {
            match self.iter.peek() {
                Some(f) if f.image.width == w && f.image.height == h => {
                    let f = self.iter.next().unwrap();
                    b.frames.extend_from_slice(&f.image.data);
                    b.meta.push((f.time, f.frame_idx));
                }
                _ => break,
            }
        }
        Some(b)
    }
}
