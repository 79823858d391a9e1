"""Host-side mirror of the reference's `matching` trait surface over the C ABI.

Same names, argument meaning and behaviour as crates/matching/src/lib.rs:7-40 and
progress.rs:3-17, implemented over include/slideo_amd.h the way
crates/matching-opencv/src/lib.rs implements it over OpenCV:

    matcher = HipImageVideoMatcher()                       # OpenCVImageVideoMatcher::default()  main.rs:69
    vm   = matcher.create_video_matcher(pages, reporter)   # lib.rs:37-64
    task = vm.match_images_with_video(video_path, rep)     # lib.rs:140-158
    matchings = task.process()                             # lib.rs:168-246

The Rust toolchain is absent from this image, so this Python mirror is what the
tests drive; the Rust shim a maintainer would add is in INTEGRATION.md.  Video
DECODE is outside the hot path (north_star starts at decoded frames; the
reference uses FFmpeg inside OpenCV videoio): `video_path` names a raw frame
container (RawVideo below) or any object with the same reader interface.
"""
import os
import struct
from dataclasses import dataclass
from typing import Any, Callable, List, Optional

import numpy as np

from . import _capi


class ProgressReporter:
    """matching::ProgressReporter (crates/matching/src/progress.rs:3-17)."""

    def __init__(self, handler: Callable[[int, int, str], None]):
        self.handler = handler

    def report(self, processed_count: int, total_count: int, message: str):
        self.handler(processed_count, total_count, message)


@dataclass
class Matching:
    """matching::Matching<I> (crates/matching/src/lib.rs:35-40)."""
    video_time: float            # seconds (std::time::Duration)
    video_frame_idx: int
    image: Optional[Any]         # Option<I>


def _load_bgr(path):
    """imread of a page image -> 8UC3 BGR (the evident intent of lib.rs:98-104)."""
    from PIL import Image
    if not os.path.exists(path):
        raise FileNotFoundError("File '%s' must exist" % path)                 # lib.rs:95-97 (panic)
    return np.ascontiguousarray(np.array(Image.open(path).convert("RGB"))[:, :, ::-1])


class RawVideo:
    """Minimal raw BGR frame container standing in for the decoder (VideoCapture, video_capture.rs:16-40).

    Layout: b'SLVF' u32 width u32 height f64 fps u64 n_frames, then n_frames * h*w*3 bytes.
    """
    MAGIC = b"SLVF"
    HDR = struct.Struct("<4sIIdQ")

    def __init__(self, path):
        self.path = path
        with open(path, "rb") as f:
            magic, self.width, self.height, self.fps, self.n_frames = self.HDR.unpack(f.read(self.HDR.size))
        if magic != self.MAGIC:
            raise ValueError("not a raw frame container: %s" % path)
        self._frame_bytes = self.width * self.height * 3

    @staticmethod
    def write(path, frames, fps):
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w, _ = frames.shape
        with open(path, "wb") as f:
            f.write(RawVideo.HDR.pack(RawVideo.MAGIC, w, h, float(fps), n))
            f.write(frames.tobytes())

    def total_frames(self):                      # CAP_PROP_FRAME_COUNT
        return float(self.n_frames)

    def total_time(self):                        # video_capture.rs:34-36
        return self.n_frames / self.fps

    def read(self, idx):
        with open(self.path, "rb") as f:
            f.seek(self.HDR.size + idx * self._frame_bytes)
            buf = f.read(self._frame_bytes)
        return np.frombuffer(buf, np.uint8).reshape(self.height, self.width, 3)


def sampled_frames(video, interval_s=5.0):
    """VideoCaptureIter (video_capture.rs:42-57): grab every frame, retrieve when
    frame_idx % floor(fps * interval) < 1; yields (frame, time_s, frame_idx)."""
    step = float(np.floor(video.fps * interval_s))
    if step <= 0:            # fps < 1 / interval: the reference's `frame_idx % 0.0` is NaN and `NaN < 1.0` is false — no frame is ever retrieved
        return
    for idx in range(int(video.n_frames)):
        if (idx % step) < 1.0:
            yield video.read(idx), idx / video.fps, idx


class HipVideoMatcherTask:
    """OpenCVVideoMatcherTask (lib.rs:161-246)."""

    def __init__(self, matcher, images, video, progress_reporter, batch=None):
        self._m, self._images, self._video, self._rep = matcher, images, video, progress_reporter
        self._batch = batch or 64 * len(getattr(matcher, "devices", [0]))       # one shard of 64 sampled frames per device and call

    def process(self) -> List[Matching]:
        video, m = self._video, self._m
        interval = 5.0
        total_time, total_frames = video.total_time(), video.total_frames()
        frames_to_process = int(total_time / interval)                                 # lib.rs:179
        results = [Matching(video_time=total_time, video_frame_idx=int(total_frames), image=None)]  # sentinel lib.rs:185-189
        name = os.path.basename(getattr(video, "path", "video"))
        progress = [0]

        def report_progress():
            progress[0] += 1
            self._rep.report(progress[0], frames_to_process, "Processing frames of '%s'..." % name)   # lib.rs:192-203

        pend_frames, pend_meta, prev_small = [], [], None

        def flush():
            nonlocal prev_small
            if not pend_frames:
                return
            stack = np.stack(pend_frames)
            changed, _, prev_small = m.changed_mask(stack, prev_small)               # MarkSimilarIter, video_capture.rs:86-98
            idx = np.nonzero(changed)[0]
            if len(idx):
                # match_images_with_frame, lib.rs:213-214 — on the copy of the frames the mask call left on the device
                verdicts = m.match_kept_frames(idx) if hasattr(m, "match_kept_frames") else m.match_frames(stack[idx])
                for j, v in zip(idx, verdicts):
                    t, fi = pend_meta[j]
                    img = self._images[v["page_idx"]] if v["page_idx"] >= 0 else None
                    results.append(Matching(video_time=t, video_frame_idx=fi, image=img))
            for _ in pend_frames:
                report_progress()
            pend_frames.clear(); pend_meta.clear()

        for frame, t, fi in sampled_frames(video, interval):
            pend_frames.append(frame); pend_meta.append((t, fi))
            if len(pend_frames) >= self._batch:
                flush()
        flush()
        self._rep.report(frames_to_process, frames_to_process, "Finished!")           # lib.rs:223-227
        return dedup_timeline(results)


def dedup_timeline(mappings: List[Matching]) -> List[Matching]:
    """lib.rs:229-244: stable sort by time, drop consecutive mappings with the same image."""
    mappings = sorted(mappings, key=lambda mm: mm.video_time)
    cleaned, last = [], None
    for mm in mappings:
        if last is not None and _same_image(last.image, mm.image):
            continue
        last = mm
        cleaned.append(mm)
    return cleaned


def _same_image(a, b):
    if a is None or b is None:
        return a is None and b is None
    return a == b


class HipVideoMatcher:
    """OpenCVVideoMatcher (lib.rs:134-158): owns the page-derived state shared by every task."""

    def __init__(self, matcher, images):
        self._m, self._images = matcher, images

    def match_images_with_video(self, video_path, progress_reporter: ProgressReporter) -> HipVideoMatcherTask:
        video = RawVideo(video_path) if isinstance(video_path, (str, os.PathLike)) else video_path
        frames_to_process = int(video.total_time() / 5.0)                             # lib.rs:148
        progress_reporter.report(0, frames_to_process, "")                            # lib.rs:150
        return HipVideoMatcherTask(self._m, self._images, video, progress_reporter)


class HipImageVideoMatcher:
    """Drop-in for OpenCVImageVideoMatcher (lib.rs:34-73) behind matching::ImageVideoMatcher."""

    def __init__(self, cfg=None, device=None, sift=None, devices=None):
        """devices: HIP ordinals, one matcher each behind one slideo_group (the reference fans out over the whole machine, the
        global rayon pool of lib.rs:45,174); None = every gfx950 device of the node; `device` = d is short for devices = [d].
        sift = (slideo_sift_config, ratio): the north-star's SIFT + L2 front end instead of the reference's ORB + Hamming
        (slideo_group_use_sift; ratio 0 = the path's own tolerance vote, > 0 = Lowe's ratio test); None = the reference's."""
        self._cfg, self._sift = cfg, sift
        self._devices = [device] if device is not None else devices

    def create_video_matcher(self, images, progress_reporter: ProgressReporter) -> HipVideoMatcher:
        """images: objects with get_path() (matching::MatchableImage, lib.rs:31-33)."""
        images = list(images)
        m = _capi.Group(self._cfg, self._devices)
        if self._sift is not None:
            m.use_sift(*self._sift)
        m.set_progress(progress_reporter.report)        # "Analyzing PDF pages..." protocol, lib.rs:43-58
        CH = 32 * len(m.devices)
        for i in range(0, len(images), CH):
            m.add_pages([_load_bgr(str(im.get_path())) for im in images[i:i + CH]])
        if not images:
            progress_reporter.report(0, 0, "Analyzing PDF pages...")
            progress_reporter.report(0, 0, "PDF page analysis successful.")
        m.set_progress(None)
        m.finalize()                                    # FlannMatcher::new, flann.rs:65-71 (raises on an empty index)
        return HipVideoMatcher(m, images)
