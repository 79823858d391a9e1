"""N > 1 path on CPU: world-size-2 gloo run of the frame sharding + the single verdict all-gather."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_frames, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from slideo_amd import distributed as D, synth
    import pyoracle as o
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pages = synth.pages(4, 800, 450, threads=1)
    frames, truth, _ = synth.frames(pages, n_frames, 640, 360, threads=1)
    lo, hi = D.shard_range(n_frames, rank, world)
    db = o.PageDB(o.default_config(nfeatures=500, min_rating=12.0))     # page DB replicated on every rank
    db.add_pages(pages, threads=1)
    assert db.finalize() == 0
    mine = db.match_frames(frames[lo:hi])                               # per-rank hot path (CPU stand-in for the GPU call)
    allv = D.all_gather_verdicts(mine, n_frames, rank, world)
    if rank == 0:
        q.put((allv.tobytes(), truth.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _mask_worker(rank, world, port, n_frames, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from slideo_amd import distributed as D, synth
    import pyoracle as o
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = _lecture_frames(synth, n_frames)
    cfg = o.default_config()
    rd, lo, hi = D.halo_range(n_frames, rank, world)
    mine = D.changed_mask_of_shard(lambda f: o.changed_mask(f, cfg), frames[rd:hi], rd < lo)
    cap = -(-n_frames // world)
    buf = torch.zeros(cap, dtype=torch.uint8); buf[: hi - lo] = torch.from_numpy(mine.astype(np.uint8))
    out = torch.empty(world * cap, dtype=torch.uint8)
    dist.all_gather_into_tensor(out, buf)
    if rank == 0:
        parts = [out[r * cap: r * cap + (D.shard_range(n_frames, r, world)[1] - D.shard_range(n_frames, r, world)[0])] for r in range(world)]
        q.put(torch.cat(parts).numpy().astype(bool).tolist())
    dist.barrier()
    dist.destroy_process_group()


def _lecture_frames(synth, n):
    """A sampled lecture: runs of the same page (later samples of a run differ by a little noise) — so that most frames
    are unchanged and the flags at the shard seam depend on the halo."""
    pages = synth.pages(3, 800, 450, threads=1)
    base, _, _ = synth.frames(pages, 3, 640, 360, threads=1)
    rng = np.random.default_rng(3)
    out = []
    for i in range(n):
        f = base[(i // 3) % 3].astype(np.int16) + rng.integers(-1, 2, base[0].shape, dtype=np.int16)
        out.append(np.clip(f, 0, 255).astype(np.uint8))
    return np.stack(out)


def test_world2_changed_mask_with_halo_equals_single_process(oracle, synth):
    """video_capture.rs:86-98 over a sharded frame list: the 1-frame halo makes the seam flags equal the unsharded ones."""
    import torch.multiprocessing as mp
    from slideo_amd import distributed as D
    n = 9
    assert D.halo_range(n, 0, 2) == (0, 0, 5) and D.halo_range(n, 1, 2) == (4, 5, 9) and D.halo_range(1, 1, 2) == (1, 1, 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_mask_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs: p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    single = oracle.changed_mask(_lecture_frames(synth, n), oracle.default_config())[0].astype(bool).tolist()
    assert got == single
    assert single[0] and not all(single) and single[5] is False          # the seam frame is unchanged: without the halo it would read "changed"


def test_shard_range_partitions():
    from slideo_amd import distributed as D
    for n in (0, 1, 7, 8, 9, 216000):
        for w in (1, 2, 3, 8):
            r = [D.shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_world2_gloo_gather_equals_single_process(oracle, synth):
    import torch.multiprocessing as mp
    from slideo_amd import distributed as D
    n = 7                                                              # odd: ragged shards (4 + 3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs: p.start()
    raw, truth = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    got = np.frombuffer(raw, dtype=oracle.VERDICT_DTYPE)
    pages = synth.pages(4, 800, 450, threads=1)
    frames, truth2, _ = synth.frames(pages, n, 640, 360, threads=1)
    db = oracle.PageDB(oracle.default_config(nfeatures=500, min_rating=12.0))
    db.add_pages(pages, threads=2)
    assert db.finalize() == 0
    single = db.match_frames(frames, threads=2)
    assert np.array_equal(got, single)
    assert got["page_idx"].tolist() == truth == truth2.tolist()
    # rank-0 post-processing: sentinel + dedup (lib.rs:185-189, 229-244)
    tl = D.timeline(got, [5.0 * i for i in range(n)], [150 * i for i in range(n)], 5.0 * n, 150 * n)
    assert tl[-1][2] == -1 or tl[-1][0] < 5.0 * n
    assert all(a[2] != b[2] for a, b in zip(tl, tl[1:]))


class _FakeMatcher:
    """Stand-in for _capi.Matcher on a CPU-only box: analyses pages with the CPU restatement and records what
    build_page_db_sharded feeds to add_page_features (the exchange is what is under test, not the analysis)."""

    def __init__(self, o, cfg):
        self.o, self.cfg, self.pages, self.imported, self.finalized = o, cfg, [], [], False

    def add_pages(self, pages):
        self.pages += [np.ascontiguousarray(p) for p in pages]

    def page_features(self, j):
        return self.o.orb(self.pages[j], self.cfg)

    def page_small(self, j):
        return self.o.small_image(self.pages[j], self.cfg.small_area)

    def add_page_features(self, w, h, kp, desc, small):
        self.imported.append((w, h, kp, desc, small))

    def finalize(self):
        self.finalized = True

    def close(self):
        pass


def _pagedb_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from slideo_amd import distributed as D, synth
    import pyoracle as o
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pages = synth.pages(5, 800, 450, threads=1)                        # 5 pages over 2 ranks: ragged (3 + 2)
    cfg = o.default_config(nfeatures=500)
    m = D.build_page_db_sharded(lambda: _FakeMatcher(o, cfg), pages, rank, world)
    if rank == 1:                                                      # the rank that analysed the smaller share
        q.put([(w, h, kp.tobytes(), desc.tobytes(), small.tobytes(), small.shape) for w, h, kp, desc, small in m.imported] if m.finalized else None)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_page_sharded_db_build_exchanges_fixed_stride_records(oracle, synth):
    """pages sharded over the ranks, ONE all_gather_into_tensor of fixed-stride records (no pickling); every rank
    ends up with every page's features in page order, bit for bit what a single process computes."""
    import torch.multiprocessing as mp
    from slideo_amd import distributed as D
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pagedb_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    pages = synth.pages(5, 800, 450, threads=1)
    cfg = oracle.default_config(nfeatures=500)
    assert got is not None and len(got) == 5
    for j, (w, h, kpb, db, sb, sshape) in enumerate(got):
        kp, desc = oracle.orb(pages[j], cfg)
        small = oracle.small_image(pages[j], cfg.small_area)
        assert (w, h) == (800, 450) and kpb == kp.tobytes() and db == desc.tobytes() and sb == small.tobytes() and tuple(sshape) == small.shape
    # the record format itself: round trip incl. an empty page and the padding slots
    recs = [(800, 450, *oracle.orb(pages[0], cfg), oracle.small_image(pages[0], cfg.small_area)),
            (640, 360, np.zeros(0, oracle.KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8), np.zeros((2, 3, 3), np.uint8))]
    buf = D.pack_page_records(recs, kp_cap=len(recs[0][2]) + 3, small_cap=recs[0][4].size + 5, n_slots=4)
    assert buf.shape[0] == 4 and buf.shape[1] % 16 == 0 and not buf[2:].any()
    for j, rec in enumerate(recs):
        w, h, kp, desc, small = D.unpack_page_record(buf[j], len(recs[0][2]) + 3)
        assert (w, h) == rec[:2] and kp.tobytes() == rec[2].tobytes() and np.array_equal(desc, rec[3]) and np.array_equal(small, rec[4])
