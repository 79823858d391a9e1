"""End-to-end run in the shape of BASELINE configs[3]/[4] on one GPU: a synthetic 2-hour lecture sampled every 5 s
(mo/lib.rs:145,175) over a deck, through the path the app drives — changed-frame mask (mo/video_capture.rs:86-98) ->
match the changed frames (mo/lib.rs:249-413) -> end-of-video sentinel + sort + consecutive-duplicate removal
(mo/lib.rs:185-189,229-244) -> videos_mapping rows (app/src/db.rs:162-191) — against the generator's ground truth.

usage (GPU box): python tools/lecture_timeline.py [--pages 200] [--hours 2] > gpurun_out/lecture.json
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slideo_amd import _capi, synth
from slideo_amd.matching import Matching, dedup_timeline
from slideo_amd.timeline import videos_mapping_rows


class Page:
    def __init__(self, nr): self.page_nr, self.pdf_hash = nr, "deck"
    def get_path(self): return "p-%d.png" % self.page_nr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=200)
    ap.add_argument("--hours", type=float, default=2.0)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    fps, interval = 30.0, 5.0
    S = int(a.hours * 3600 / interval)                       # sampled frames
    rng = np.random.default_rng(0xF4A3E5)
    pages = synth.pages(a.pages, 2001, 1125, threads=64)
    # visits: geometric dwell, mean 12 samples (~1 page change per minute); every tenth visit shows no slide
    visits, s = [], 0
    while s < S:
        d = int(min(rng.geometric(1 / 12.0), S - s))
        visits.append((s, d, -1 if rng.random() < 0.1 else int(rng.integers(0, a.pages))))
        s += d
    truth_of_sample = np.concatenate([np.full(d, p) for _, d, p in visits])

    m = _capi.Matcher(_capi.default_config(nfeatures=1000))
    t0 = time.time()
    for i in range(0, a.pages, 50):
        m.add_pages(list(pages[i:i + 50]))
    m.finalize()
    t_db = time.time() - t0
    imgs = [Page(i + 1) for i in range(a.pages)]

    def sample_frames(lo, hi):
        out = np.empty((hi - lo, 1080, 1920, 3), np.uint8)
        for v, (s0, d, p) in enumerate(visits):
            a0, a1 = max(s0, lo), min(s0 + d, hi)
            if a0 >= a1: continue
            if p >= 0:
                base, tp, _ = synth.frames(pages[p:p + 1], 1, 1920, 1080, first=1000 + 2 * v, threads=1)
                k = 0
                while tp[0] < 0:                              # the generator's own "no slide" draw: take the next seed
                    k += 1
                    base, tp, _ = synth.frames(pages[p:p + 1], 1, 1920, 1080, first=100000 + 97 * v + k, threads=1)
            else:
                base = rng.integers(0, 40, (1, 1080, 1920, 3), dtype=np.uint8)       # dark noisy scene
            for sidx in range(a0, a1):                        # later samples of a visit: the same picture, fresh sensor noise
                n = np.random.default_rng(sidx).integers(-2, 3, base[0].shape, dtype=np.int16)
                out[sidx - lo] = np.clip(base[0].astype(np.int16) + (n if sidx > s0 else 0), 0, 255).astype(np.uint8)
        return out

    results = [Matching(video_time=S * interval, video_frame_idx=int(S * interval * fps), image=None)]    # sentinel
    prev_small, n_changed, t_gpu, t_gen = None, 0, 0.0, 0.0
    for lo in range(0, S, a.batch):
        hi = min(S, lo + a.batch)
        t0 = time.time(); stack = sample_frames(lo, hi); t_gen += time.time() - t0
        t0 = time.time()
        changed, _, prev_small = m.changed_mask(stack, prev_small)
        idx = np.nonzero(changed)[0]
        if len(idx):
            v = m.match_frames(stack[idx])
            for j, r in zip(idx, v):
                sidx = lo + int(j)
                results.append(Matching(video_time=sidx * interval, video_frame_idx=int(sidx * interval * fps),
                                        image=imgs[r["page_idx"]] if r["page_idx"] >= 0 else None))
        n_changed += len(idx)
        t_gpu += time.time() - t0
    tl = dedup_timeline(results)
    rows = videos_mapping_rows(tl)
    # ground truth through the same sentinel + dedup
    tr = [Matching(video_time=S * interval, video_frame_idx=0, image=None)]
    tr += [Matching(video_time=s0 * interval, video_frame_idx=0, image=imgs[p] if p >= 0 else None) for s0, d, p in visits]
    tt = dedup_timeline(tr)
    key = lambda mm: (round(mm.video_time, 3), mm.image.page_nr if mm.image else 0)
    got, want = set(map(key, tl)), set(map(key, tt))
    out = {"workload": "synthetic lecture, %.1f h, sampled every 5 s = %d frames of 1920x1080, %d-page deck (2001x1125), ORB-1000" % (a.hours, S, a.pages),
           "sampled_frames": S, "visits": len(visits), "changed_frames_matched": int(n_changed),
           "changed_fraction": round(n_changed / S, 4),
           "timeline_entries": len(tl), "truth_entries": len(tt), "entries_equal_to_truth": len(got & want),
           "missing": len(want - got), "spurious": len(got - want), "videos_mapping_rows": len(rows),
           "page_db_build_s": round(t_db, 2), "gpu_path_s_incl_h2d": round(t_gpu, 2), "frame_synthesis_s": round(t_gen, 1),
           "sampled_frames_per_s_incl_h2d": round(S / t_gpu, 1),
           "note": "host frames in, mask + match + timeline out; the mask call and the H2D copies are inside gpu_path_s"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
