"""Host-side mirror of crates/matching (slideo_amd/matching.py): CPU logic here, GPU end-to-end below."""
import os

import numpy as np
import pytest

from slideo_amd import matching as mt


def test_dedup_timeline_matches_reference_rule(oracle):
    # lib.rs:229-244 — stable sort by time then drop consecutive equal images; None compares equal to None
    ms = [mt.Matching(25.0, 750, None), mt.Matching(0.0, 0, "p3"), mt.Matching(5.0, 150, "p3"), mt.Matching(10.0, 300, None),
          mt.Matching(15.0, 450, None), mt.Matching(20.0, 600, "p4")]
    out = mt.dedup_timeline(ms)
    assert [(m.video_time, m.image) for m in out] == [(0.0, "p3"), (10.0, None), (20.0, "p4"), (25.0, None)]
    # same rule in the oracle
    pages = {"p3": 3, "p4": 4, None: -1}
    keep = oracle.timeline_dedup([int(m.video_time * 1000) for m in ms], [pages[m.image] for m in ms])
    assert [ms[i].video_time for i in keep] == [m.video_time for m in out]


def test_raw_video_and_sampler(tmp_path):
    frames = np.random.default_rng(0).integers(0, 256, (40, 6, 8, 3), dtype=np.uint8)
    p = os.path.join(tmp_path, "v.slvf")
    mt.RawVideo.write(p, frames, fps=2.5)
    v = mt.RawVideo(p)
    assert (v.width, v.height, v.n_frames, v.fps) == (8, 6, 40, 2.5)
    assert v.total_time() == 16.0
    got = list(mt.sampled_frames(v, 5.0))
    # video_capture.rs:52: frame_idx % floor(fps*5) < 1  with floor(12.5) = 12
    assert [g[2] for g in got] == [0, 12, 24, 36]
    assert [g[1] for g in got] == [0.0, 4.8, 9.6, 14.4]
    assert np.array_equal(got[1][0], frames[12])


def test_progress_reporter_contract():
    seen = []
    r = mt.ProgressReporter(lambda a, b, c: seen.append((a, b, c)))
    r.report(1, 2, "x")
    assert seen == [(1, 2, "x")]


class Page:
    def __init__(self, path, nr):
        self.path, self.page_nr = path, nr

    def get_path(self):
        return self.path

    def __eq__(self, o):
        return isinstance(o, Page) and o.page_nr == self.page_nr


@pytest.mark.gpu
def test_trait_surface_end_to_end(tmp_path, capi, synth):
    """create_video_matcher -> match_images_with_video -> process, as crates/app/src/main.rs:69-93 drives it."""
    from PIL import Image
    pages = synth.pages(4, 800, 450)
    page_objs = []
    for i, p in enumerate(pages):
        path = os.path.join(tmp_path, "p-%d.png" % (i + 1))              # poppler naming, pdftocairo.rs:217-226
        Image.fromarray(p[:, :, ::-1]).save(path)
        page_objs.append(Page(path, i + 1))
    frames, truth, _ = synth.frames(pages, 6, 640, 360)
    # a 1 fps "video": every page shown for 10 s -> samples every 5 s see each frame twice
    seq = np.repeat(frames, 10, axis=0)
    vid = os.path.join(tmp_path, "v.slvf")
    mt.RawVideo.write(vid, seq, fps=1.0)
    log = []
    rep = mt.ProgressReporter(lambda a, b, c: log.append((a, b, c)))
    cfg = capi.default_config(nfeatures=500, min_rating=12.0)
    vm = mt.HipImageVideoMatcher(cfg).create_video_matcher(page_objs, rep)
    assert log[0] == (0, 4, "Analyzing PDF pages...") and log[-1] == (4, 4, "PDF page analysis successful.")   # lib.rs:43-58
    task = vm.match_images_with_video(vid, rep)
    assert log[-1] == (0, 12, "")                                                                            # lib.rs:148-150
    out = task.process()
    assert log[-1] == (12, 12, "Finished!")
    # expected timeline from the ground truth: one entry per change of page (+ end sentinel)
    exp = []
    for i, t in enumerate(truth):
        pg = None if t < 0 else page_objs[t]
        if not exp or not mt._same_image(exp[-1][1], pg):
            exp.append((i * 10.0, pg))
    if exp[-1][1] is not None:
        exp.append((60.0, None))
    assert [(m.video_time, m.image) for m in out] == exp
    assert all(isinstance(m, mt.Matching) for m in out)


@pytest.mark.gpu
def test_cpp_host_mirror_matches_python_mirror(tmp_path, capi, synth):
    """slideo_amd/host/matching.hpp (compiled, g++) drives the same C ABI and yields the same timeline."""
    import subprocess
    from slideo_amd import build
    exe = build.build_host_demo()
    pages = synth.pages(4, 800, 450)
    # a pdftocairo target directory: p-<nr>.png; the C++ mirror scans and decodes it itself
    from PIL import Image
    lst = os.path.join(tmp_path, "pages")
    os.makedirs(lst)
    for i, p in enumerate(pages):
        Image.fromarray(np.ascontiguousarray(p[:, :, ::-1])).save(os.path.join(lst, "p-%d.png" % (i + 1)))
    frames, truth, _ = synth.frames(pages, 6, 640, 360)
    vid = os.path.join(tmp_path, "v.slvf")
    mt.RawVideo.write(vid, np.repeat(frames, 10, axis=0), fps=1.0)
    out = subprocess.run([exe, lst, vid, "500", "12", "vidhash"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    got = [tuple(int(x) for x in ln.split()) for ln in lines if ln[0].isdigit()]
    exp = []
    for i, t in enumerate(truth):
        nr = 0 if t < 0 else int(t) + 1
        if not exp or exp[-1][1] != nr:
            exp.append((i * 10000, nr))
    if exp[-1][1] != 0:
        exp.append((60000, 0))
    assert got == exp
    # output contract: the C++ rows / viewer records equal the Python restatement's
    from slideo_amd import timeline as tl

    class Pg:
        def __init__(self, nr): self.page_nr, self.pdf_hash = nr, "pdfhash"
    ms = [mt.Matching(video_time=t / 1000.0, video_frame_idx=0, image=None if nr == 0 else Pg(nr)) for t, nr in exp]
    rows = tl.videos_mapping_rows(ms)
    assert [ln for ln in lines if ln.startswith("row ")] == ["row %d %s %d" % (r.video_ms, r.pdf_hash or "-", r.page) for r in rows]
    assert [ln for ln in lines if ln.startswith("pvm ")] == ["pvm %d %s %s %d %d" % (p.video_offset_ms, p.pdf_hash, p.video_hash, p.page_idx, p.duration_ms)
                                                           for p in tl.pdf_video_matchings(rows, "pdfhash", "vidhash")]
