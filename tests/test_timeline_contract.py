"""videos_mapping rows and PdfVideoMatching records (crates/app/src/db.rs:162-260), hand-computed."""
from dataclasses import dataclass

from slideo_amd import timeline as tl
from slideo_amd.matching import Matching, dedup_timeline


@dataclass(frozen=True)
class Page:
    pdf_hash: str
    page_nr: int


A1, A2, B7 = Page("aaa", 1), Page("aaa", 2), Page("bbb", 7)


def test_rows_follow_db_rs():
    ms = [Matching(0.0, 0, A1), Matching(5.0009, 150, None), Matching(10.0, 300, B7), Matching(61.9999, 1860, None)]
    rows = tl.videos_mapping_rows(ms)
    # as_millis() truncates; page = page_nr - 1; None -> NULL hash, page 0
    assert rows == [tl.VideoMappingRow(0, "aaa", 0), tl.VideoMappingRow(5000, None, 0),
                    tl.VideoMappingRow(10000, "bbb", 6), tl.VideoMappingRow(61999, None, 0)]


def test_viewer_records_durations_and_filter():
    rows = [tl.VideoMappingRow(10000, "bbb", 6), tl.VideoMappingRow(0, "aaa", 0), tl.VideoMappingRow(5000, None, 0),
            tl.VideoMappingRow(25000, "aaa", 1), tl.VideoMappingRow(60000, None, 0)]
    got = tl.pdf_video_matchings(rows, "aaa", "vid")
    # ORDER BY video_ms; duration = next.video_ms - this.video_ms over ALL mappings of the video; only `aaa` rows reported
    assert got == [tl.PdfVideoMatching(0, "aaa", "vid", 0, 5000), tl.PdfVideoMatching(25000, "aaa", "vid", 1, 35000)]
    assert tl.pdf_video_matchings(rows, "bbb", "vid") == [tl.PdfVideoMatching(10000, "bbb", "vid", 6, 15000)]
    assert tl.pdf_video_matchings(rows, "zzz", "vid") == []


def test_last_mapping_without_successor_gets_5000():
    rows = [tl.VideoMappingRow(0, "aaa", 0)]                     # db.rs:243-245 ("should not happen anymore")
    assert tl.pdf_video_matchings(rows, "aaa", "v") == [tl.PdfVideoMatching(0, "aaa", "v", 0, 5000)]


def test_from_matcher_timeline_end_to_end():
    # sampled verdicts every 5 s: A1 A1 None A2 A2, then the end-of-video sentinel (lib.rs:185-189) at 27.3 s
    raw = [Matching(27.3, 819, None), Matching(0.0, 0, A1), Matching(5.0, 150, A1), Matching(10.0, 300, None),
           Matching(15.0, 450, A2), Matching(20.0, 600, A2)]
    rows = tl.videos_mapping_rows(dedup_timeline(raw))
    assert [(r.video_ms, r.pdf_hash, r.page) for r in rows] == [(0, "aaa", 0), (10000, None, 0), (15000, "aaa", 1), (27300, None, 0)]
    assert [(p.video_offset_ms, p.page_idx, p.duration_ms) for p in tl.pdf_video_matchings(rows, "aaa", "v")] == [(0, 0, 10000), (15000, 1, 12300)]
