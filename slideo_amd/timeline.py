"""Timeline output contract of the reference's app crate, restated for the drop-in.

The matcher returns `Vec<Matching<I>>` (crates/matching/src/lib.rs:35-40); crates/app turns it into
  * rows of the `videos_mapping(video_id, video_ms, pdf_hash, page)` table
    (crates/app/src/db.rs:162-191, schema migrations/20210309093718_setup.sql:22-29), and
  * `PdfVideoMatching{video_offset_ms, pdf_hash, video_hash, page_idx, duration_ms}` records served to the
    viewer (db.rs:194-201, 209-260).
These two pure functions reproduce both so that a caller of this framework can feed the existing viewer
unchanged.  Host logic only — nothing here touches the GPU.
"""
from dataclasses import dataclass
from typing import Iterable, List, Optional

U32 = 0xFFFFFFFF


@dataclass(frozen=True)
class VideoMappingRow:                  # one INSERT of db.rs:179-186
    video_ms: int                       # matching.video_time.as_millis() as u32   (db.rs:176)
    pdf_hash: Optional[str]             # matching.image.map(|p| p.pdf_hash)        (db.rs:175)
    page: int                           # (page_nr - 1) as u32, 0 for None          (db.rs:177)


@dataclass(frozen=True)
class PdfVideoMatching:                 # db.rs:194-201
    video_offset_ms: int
    pdf_hash: str
    video_hash: str
    page_idx: int
    duration_ms: int


def videos_mapping_rows(matchings: Iterable) -> List[VideoMappingRow]:
    """db.rs:174-188.  `matchings`: slideo_amd.matching.Matching whose .image (or None) has .pdf_hash and .page_nr
    (1-based, pdf_to_images.rs:19-31).  Order is preserved (the caller passes the de-duplicated timeline)."""
    rows = []
    for m in matchings:
        img = m.image
        ms = int(m.video_time * 1000.0) & U32            # Duration::as_millis() truncates; `as u32` wraps
        rows.append(VideoMappingRow(video_ms=ms, pdf_hash=None if img is None else img.pdf_hash,
                                    page=0 if img is None else (int(img.page_nr) - 1) & U32))
    return rows


def pdf_video_matchings(rows: Iterable[VideoMappingRow], pdf_hash: str, video_hash: str) -> List[PdfVideoMatching]:
    """db.rs:224-258 for the mappings of ONE video: ORDER BY video_ms ASC (stable), duration = distance to the next
    mapping (5000 for the last one — "should not happen anymore": the matcher always appends the end-of-video
    sentinel, lib.rs:185-189), and only the mappings that show a page of `pdf_hash` are reported."""
    ordered = sorted(rows, key=lambda r: r.video_ms)
    out = []
    for i, r in enumerate(ordered):
        duration = (ordered[i + 1].video_ms - r.video_ms) if i + 1 < len(ordered) else 5000
        if r.pdf_hash is not None and r.pdf_hash == pdf_hash:
            out.append(PdfVideoMatching(video_offset_ms=r.video_ms & U32, pdf_hash=r.pdf_hash, video_hash=video_hash,
                                        page_idx=r.page & U32, duration_ms=duration & U32))
    return out
