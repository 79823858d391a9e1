"""End-to-end run in the shape of BASELINE configs[3]/[4]: a synthetic lecture sampled every 5 s (mo/lib.rs:145,175) over a
deck, through the path the app drives — changed-frame mask (mo/video_capture.rs:86-98) -> match the changed frames
(mo/lib.rs:249-413) -> end-of-video sentinel + sort + consecutive-duplicate removal (mo/lib.rs:185-189,229-244) ->
videos_mapping rows (app/src/db.rs:162-191) — against the generator's ground truth.

One process, or sharded: rank r of `world` takes a contiguous block of the sampled frames plus the one frame before it
(the changed-frame test compares with the previous SAMPLED frame: slideo_amd.distributed.halo_range), every rank holds
the whole page DB, and the ranks exchange ONE all-gather of per-frame records.  tests/test_gpu_big_shapes.py
runs the sharded form with two ranks that both drive the HIP library on one GPU (gloo) and compares with the one-process run.

usage (GPU box): python tools/lecture_timeline.py [--pages 200] [--hours 2] [--devices 0,0] > gpurun_out/lecture.json
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slideo_amd import _capi, synth, distributed as D
from slideo_amd.matching import Matching, dedup_timeline
from slideo_amd.timeline import videos_mapping_rows

FPS, INTERVAL = 30.0, 5.0


class Page:
    def __init__(self, nr): self.page_nr, self.pdf_hash = nr, "deck"
    def get_path(self): return "p-%d.png" % self.page_nr
    def __eq__(self, o): return isinstance(o, Page) and o.page_nr == self.page_nr
    def __hash__(self): return hash(self.page_nr)


def make_visits(n_samples, n_pages, seed=0xF4A3E5):
    """visits (first sample, samples, page or -1): geometric dwell, mean 12 samples (~1 page change per minute); every
    tenth visit shows no slide"""
    rng = np.random.default_rng(seed)
    visits, s = [], 0
    while s < n_samples:
        d = int(min(rng.geometric(1 / 12.0), n_samples - s))
        visits.append((s, d, -1 if rng.random() < 0.1 else int(rng.integers(0, n_pages))))
        s += d
    return visits


def sample_frames(pages, visits, lo, hi, w=1920, h=1080):
    """sampled frames [lo, hi) of the lecture: a pure function of (pages, visits, sample index), so that every rank of a
    sharded run synthesises exactly the frames a single process would"""
    out = np.empty((hi - lo, h, w, 3), np.uint8)
    for v, (s0, d, p) in enumerate(visits):
        a0, a1 = max(s0, lo), min(s0 + d, hi)
        if a0 >= a1:
            continue
        if p >= 0:
            base, tp, _ = synth.frames(pages[p:p + 1], 1, w, h, first=1000 + 2 * v, threads=1)
            k = 0
            while tp[0] < 0:                              # the generator's own "no slide" draw: take the next seed
                k += 1
                base, tp, _ = synth.frames(pages[p:p + 1], 1, w, h, first=100000 + 97 * v + k, threads=1)
        else:
            base = np.random.default_rng(7919 * v + 13).integers(0, 40, (1, h, w, 3), dtype=np.uint8)       # dark noisy scene
        for sidx in range(a0, a1):                        # later samples of a visit: the same picture, fresh sensor noise
            n = np.random.default_rng(sidx).integers(-2, 3, base[0].shape, dtype=np.int16)
            out[sidx - lo] = np.clip(base[0].astype(np.int16) + (n if sidx > s0 else 0), 0, 255).astype(np.uint8)
    return out


def truth_timeline(visits, n_samples, imgs):
    tr = [Matching(video_time=n_samples * INTERVAL, video_frame_idx=0, image=None)]
    tr += [Matching(video_time=s0 * INTERVAL, video_frame_idx=0, image=imgs[p] if p >= 0 else None) for s0, d, p in visits]
    return dedup_timeline(tr)


def timeline_key(mm):
    return (round(mm.video_time, 3), mm.image.page_nr if mm.image else 0)


def run_shard(m, pages, visits, n_samples, rank=0, world=1, batch=256, w=1920, h=1080):
    """This rank's block of the lecture through the HIP library: (changed flags, page_idx per sample of the block;
    -2 where the sample was not matched because it is unchanged), plus timings."""
    rd, lo, hi = D.halo_range(n_samples, rank, world)
    changed_all = np.zeros(hi - lo, bool)
    page_of = np.full(hi - lo, -2, np.int32)
    prev_small, t_gpu, t_gen = None, 0.0, 0.0
    if rd < lo:                                            # the halo frame only provides the small image sample `lo` is compared with
        t0 = time.time(); halo = sample_frames(pages, visits, rd, lo, w, h); t_gen += time.time() - t0
        t0 = time.time(); _, _, prev_small = m.changed_mask(halo, None); t_gpu += time.time() - t0
    for a in range(lo, hi, batch):
        b = min(hi, a + batch)
        t0 = time.time(); stack = sample_frames(pages, visits, a, b, w, h); t_gen += time.time() - t0
        t0 = time.time()
        changed, _, prev_small = m.changed_mask(stack, prev_small)
        changed_all[a - lo:b - lo] = changed
        idx = np.nonzero(changed)[0]
        if len(idx):
            v = m.match_kept_frames(idx)                   # the mask call's upload, matched in place (no second H2D copy)
            page_of[a - lo + idx] = v["page_idx"]
        t_gpu += time.time() - t0
    return changed_all, page_of, t_gpu, t_gen


def timeline_from_samples(page_of, n_samples, imgs):
    """sentinel + the changed samples' verdicts -> sort + consecutive-duplicate removal (mo/lib.rs:185-189,229-244)"""
    results = [Matching(video_time=n_samples * INTERVAL, video_frame_idx=int(n_samples * INTERVAL * FPS), image=None)]
    for sidx, p in enumerate(page_of):
        if p != -2:
            results.append(Matching(video_time=sidx * INTERVAL, video_frame_idx=int(sidx * INTERVAL * FPS), image=imgs[p] if p >= 0 else None))
    return dedup_timeline(results)


def build_matcher(pages, nfeatures=1000, device=0, devices=None):
    """one matcher, or — `devices` given — the N-device group of include/slideo_amd.h (same calls, frames sharded inside)"""
    cfg = _capi.default_config(nfeatures=nfeatures)
    m = _capi.Group(cfg, devices=devices) if devices else _capi.Matcher(cfg, device=device)
    for i in range(0, len(pages), 50):
        m.add_pages(list(pages[i:i + 50]))
    m.finalize()
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=200)
    ap.add_argument("--hours", type=float, default=2.0)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--devices", default="", help="comma-separated ordinals: run through slideo_group_* (an ordinal may repeat)")
    a = ap.parse_args()
    devices = [int(x) for x in a.devices.split(",") if x != ""]
    S = int(a.hours * 3600 / INTERVAL)                       # sampled frames
    pages = synth.pages(a.pages, 2001, 1125, threads=64)
    visits = make_visits(S, a.pages)
    t0 = time.time()
    m = build_matcher(pages, devices=devices)
    t_db = time.time() - t0
    imgs = [Page(i + 1) for i in range(a.pages)]
    changed, page_of, t_gpu, t_gen = run_shard(m, pages, visits, S, 0, 1, a.batch)
    tl = timeline_from_samples(page_of, S, imgs)
    rows = videos_mapping_rows(tl)
    tt = truth_timeline(visits, S, imgs)
    got, want = set(map(timeline_key, tl)), set(map(timeline_key, tt))
    out = {"workload": "synthetic lecture, %.1f h, sampled every 5 s = %d frames of 1920x1080, %d-page deck (2001x1125), ORB-1000" % (a.hours, S, a.pages),
           "sampled_frames": S, "visits": len(visits), "changed_frames_matched": int(changed.sum()),
           "changed_fraction": round(float(changed.mean()), 4),
           "timeline_entries": len(tl), "truth_entries": len(tt), "entries_equal_to_truth": len(got & want),
           "missing": len(want - got), "spurious": len(got - want), "videos_mapping_rows": len(rows),
           "page_db_build_s": round(t_db, 2), "gpu_path_s_incl_h2d": round(t_gpu, 2), "frame_synthesis_s": round(t_gen, 1),
           "sampled_frames_per_s_incl_h2d": round(S / t_gpu, 1),
           "devices": devices or [0], "boundary": "slideo_group_*" if devices else "slideo_matcher_*",
           "note": "host frames in, mask + match + timeline out; the mask call and the H2D copies are inside gpu_path_s"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
