//! Host-side decode helpers: page PNGs and sampled video frames, through the OpenCV modules the application already
//! links (imgcodecs, videoio).  No matching logic lives here.
//!
//! The sampling RULE is the reference's (crates/matching-opencv/src/video_capture.rs:42-57) and lives in `SamplingRule`; the
//! decoder behind it is a trait (`FrameSource`), with OpenCV's VideoCapture as the one implementation shipped.
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

// ---- sampling ------------------------------------------------------------------------------------------------------
// Three separate pieces, so that the decoder is replaceable (a VCN / FFmpeg-direct source, a test source) without touching
// the rule that decides WHICH frames the matcher sees — the one thing here that must equal the reference:
//   FrameSource     what a decoder has to offer: position, rate, length, "skip one frame", "decode the frame just skipped to"
//   SamplingRule    the reference's rule (crates/matching-opencv/src/video_capture.rs:42-57): the frame at position p is sampled
//                   iff p mod floor(fps * interval) < 1, and its time is p / fps
//   Sampler         the iterator: walks a source frame by frame and decodes only what the rule keeps
//   OpenCvSource    the FrameSource over opencv::videoio::VideoCapture (what the reference's VideoCaptureIter wraps)

/// A frame-accurate sequential decoder.
pub trait FrameSource {
    /// frames per second of the stream (video_capture.rs:22)
    fn fps(&self) -> f64;
    /// frames in the stream as the container reports them (video_capture.rs:31)
    fn frame_count(&self) -> f64;
    /// position of the NEXT frame, counted in frames (video_capture.rs:45)
    fn position(&self) -> f64;
    /// moves past the next frame without decoding it; false at the end of the stream (video_capture.rs:48)
    fn advance(&mut self) -> bool;
    /// decodes the frame `advance` has just moved past (video_capture.rs:53)
    fn decode_current(&mut self) -> BgrImage;
}

/// Which positions are sampled, and what time they carry.
#[derive(Clone, Copy)]
pub struct SamplingRule {
    /// floor(fps * interval): the period in frames (video_capture.rs:52)
    period: f64,
    fps: f64,
}

impl SamplingRule {
    pub fn new(fps: f64, interval: Duration) -> Self {
        SamplingRule { period: (fps * interval.as_secs_f64()).floor(), fps }
    }
    /// video_capture.rs:52 (f64 remainder, so a fractional position reported by a container still compares as there)
    pub fn keeps(&self, position: f64) -> bool {
        position % self.period < 1.0
    }
    /// video_capture.rs:46
    pub fn time_of(&self, position: f64) -> Duration {
        Duration::from_secs_f64(position / self.fps)
    }
    /// video_capture.rs:35
    pub fn total_time(&self, frame_count: f64) -> Duration {
        Duration::from_secs_f64(frame_count / self.fps)
    }
}

pub struct SampledFrame {
    pub image: BgrImage,
    pub time: Duration,
    pub frame_idx: usize,
}

/// The sampled frames of a source, in order.
pub struct Sampler<S: FrameSource> {
    source: S,
    rule: SamplingRule,
}

impl<S: FrameSource> Sampler<S> {
    pub fn new(source: S, interval: Duration) -> Self {
        let rule = SamplingRule::new(source.fps(), interval);
        Sampler { source, rule }
    }
    pub fn total_frames(&self) -> f64 {
        self.source.frame_count()
    }
    pub fn total_time(&self) -> Duration {
        self.rule.total_time(self.source.frame_count())
    }
}

impl<S: FrameSource> Iterator for Sampler<S> {
    type Item = SampledFrame;

    fn next(&mut self) -> Option<SampledFrame> {
        // every frame is moved past (the reference grabs every frame); only the kept ones are decoded
        loop {
            let at = self.source.position();
            if !self.source.advance() {
                return None;
            }
            if self.rule.keeps(at) {
                let image = self.source.decode_current();
                return Some(SampledFrame { image, time: self.rule.time_of(at), frame_idx: at as usize });
            }
        }
    }
}

/// FrameSource over OpenCV's VideoCapture (FFmpeg inside videoio) — the decoder the application already links.
pub struct OpenCvSource {
    video: VideoCapture,
    scratch: Mat,
}

impl OpenCvSource {
    /// video_capture.rs:17-21 (backend 0 = any)
    pub fn open(path: &Path) -> Self {
        OpenCvSource { video: VideoCapture::from_file(&path.to_string_lossy(), 0).unwrap(), scratch: Mat::default() }
    }
}

impl FrameSource for OpenCvSource {
    fn fps(&self) -> f64 {
        self.video.get(CAP_PROP_FPS).unwrap()
    }
    fn frame_count(&self) -> f64 {
        self.video.get(CAP_PROP_FRAME_COUNT).unwrap()
    }
    fn position(&self) -> f64 {
        self.video.get(CAP_PROP_POS_FRAMES).unwrap()
    }
    fn advance(&mut self) -> bool {
        self.video.grab().unwrap()
    }
    fn decode_current(&mut self) -> BgrImage {
        self.video.retrieve(&mut self.scratch, 0).unwrap();
        mat_to_bgr(&self.scratch)
    }
}

/// The sampled frames of a video file, decoded by OpenCV.
pub fn sampled_video(path: &Path, interval: Duration) -> Sampler<OpenCvSource> {
    Sampler::new(OpenCvSource::open(path), interval)
}

#[cfg(test)]
mod tests {
    use super::*;

    /// 100 synthetic frames at 30 fps, 1 s interval: the rule keeps every 30th position and nothing else is decoded.
    struct Counting {
        at: f64,
        n: f64,
        decoded: Vec<usize>,
    }
    impl FrameSource for Counting {
        fn fps(&self) -> f64 {
            30.0
        }
        fn frame_count(&self) -> f64 {
            self.n
        }
        fn position(&self) -> f64 {
            self.at
        }
        fn advance(&mut self) -> bool {
            if self.at >= self.n {
                return false;
            }
            self.at += 1.0;
            true
        }
        fn decode_current(&mut self) -> BgrImage {
            self.decoded.push(self.at as usize - 1);
            BgrImage { data: vec![0; 3], width: 1, height: 1 }
        }
    }

    #[test]
    fn samples_every_period_and_decodes_nothing_else() {
        let s = Sampler::new(Counting { at: 0.0, n: 100.0, decoded: vec![] }, Duration::from_secs(1));
        assert_eq!(s.total_time(), Duration::from_secs_f64(100.0 / 30.0));
        let got: Vec<(usize, Duration)> = s.map(|f| (f.frame_idx, f.time)).collect();
        assert_eq!(got.iter().map(|g| g.0).collect::<Vec<_>>(), vec![0, 30, 60, 90]);
        assert_eq!(got[2].1, Duration::from_secs(2));
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
