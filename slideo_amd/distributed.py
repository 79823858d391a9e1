"""Multi-GPU sharding of the frame path: one process per GPU.

Frames are independent once the page DB exists (crates/matching-opencv/src/lib.rs:213-214), so
rank r takes a contiguous block of the sampled frames, the page DB is replicated, and the only
exchange is ONE all-gather of fixed-size verdict records; rank 0 then runs the reference's
sort + consecutive-duplicate removal (lib.rs:229-244).  Backend "nccl" is RCCL on ROCm (GPU
tensors); "gloo" works on CPU tensors and is what the CPU-only tests use.

(An RCCL communicator initialised BEFORE the matcher used to cost every rank 10 %: its streams shifted the matcher's slot streams
onto shared hardware queues.  slideo_matcher_create now picks its slot streams by measurement — include/slideo_amd.h "Environment".)
"""
import numpy as np

VERDICT_WORDS = 4   # page_idx i32, similarity f32 (bit pattern), inliers i32, n_keypoints i32


def shard_range(n_frames, rank, world):
    """Contiguous block [lo, hi) of rank `rank` (block sizes differ by at most one)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def halo_range(n_frames, rank, world):
    """Frames a rank must READ for the changed-frame test of its block: the block plus the one frame before it.

    MarkSimilarIter (crates/matching-opencv/src/video_capture.rs:86-98) compares every sampled frame with the previous
    sampled frame (the reference updates `last` on every sample, :97) and always keeps the first of the video (:92), so
    a contiguous shard needs exactly a 1-frame halo at its start; rank 0 has none.  Returns (lo_read, lo, hi)."""
    lo, hi = shard_range(n_frames, rank, world)
    return (lo - 1 if lo > 0 and hi > lo else lo), lo, hi


def changed_mask_of_shard(changed_mask_fn, frames_read, has_halo):
    """`changed` flags of a shard given the frames of halo_range() and the single-process mask function
    (`Matcher.changed_mask` on the GPU, the CPU restatement in the tests): the halo frame only provides the small image
    the first real frame is compared with; its own flag (always "changed", it is first in the call) is dropped."""
    if len(frames_read) == 0:
        return np.zeros(0, bool)
    changed = np.asarray(changed_mask_fn(frames_read)[0], bool)
    return changed[1:] if has_halo else changed


def all_gather_verdicts(verdicts, n_total, rank, world, device=None):
    """The one collective of the frame path: fixed-size verdict records (16 B per frame), padded to the largest shard.

    verdicts: this rank's block, either a structured array (_capi.VERDICT_DTYPE, host) or an int32 torch tensor [n, 4]
    that already lives on the collective's device — what slideo_match_frames_collect_dev leaves in device memory — in
    which case nothing passes through the host before the all-gather (bench.py does the same with a preallocated tensor).
    Returns the concatenation over ranks in frame order (length n_total, structured array) on every rank."""
    import torch
    import torch.distributed as dist
    from ._capi import VERDICT_DTYPE
    is_tensor = isinstance(verdicts, torch.Tensor)
    if world == 1:
        return (verdicts.cpu().numpy().view(VERDICT_DTYPE).reshape(-1) if is_tensor else verdicts.copy())
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    if is_tensor:
        t = torch.zeros((cap, VERDICT_WORDS), dtype=torch.int32, device=verdicts.device)
        t[: verdicts.shape[0]] = verdicts.reshape(-1, VERDICT_WORDS)
    else:
        buf = np.zeros((cap, VERDICT_WORDS), np.int32)
        buf[: len(verdicts)] = verdicts.view(np.int32).reshape(-1, VERDICT_WORDS)
        t = torch.from_numpy(buf)
        if device is not None:
            t = t.to(device)
    out = torch.empty((world * cap, VERDICT_WORDS), dtype=torch.int32, device=t.device)
    dist.all_gather_into_tensor(out, t)          # the single collective of the path
    out = out.cpu().numpy().reshape(world, cap, VERDICT_WORDS)
    parts = [out[r, : hi - lo] for r, (lo, hi) in enumerate(sizes)]
    return np.ascontiguousarray(np.concatenate(parts)).view(VERDICT_DTYPE).reshape(-1)


def timeline(verdicts, times_s, frame_idx, total_time_s, total_frames, changed=None):
    """lib.rs:185-189 + 229-244 on gathered verdicts: end-of-video sentinel, stable sort by time,
    drop consecutive equal pages.  Returns a list of (time_s, frame_idx, page_idx or -1).

    The reference pushes a Matching only for CHANGED frames (lib.rs:205-208): pass the gathered `changed` flags when
    `verdicts` covers every sampled frame (an unchanged frame's record is not a verdict and would break the
    consecutive-duplicate removal); None = the verdicts are already those of the changed frames only."""
    rows = [(float(total_time_s), int(total_frames), -1)]
    keep = np.ones(len(verdicts), bool) if changed is None else np.asarray(changed, bool)
    if len(keep) != len(verdicts):
        raise ValueError("`changed` must have one flag per verdict")
    rows += [(float(t), int(i), int(p)) for t, i, p, k in zip(times_s, frame_idx, verdicts["page_idx"], keep) if k]
    rows.sort(key=lambda r: r[0])
    out, last = [], None
    for r in rows:
        if last is not None and last == r[2]:
            continue
        last = r[2]
        out.append(r)
    return out


PAGE_REC_HEADER = 8      # int32 words: width, height, n_keypoints, small_w, small_h, 3 spare


def pack_page_records(recs, kp_cap, small_cap, n_slots):
    """Fixed-stride page records for the all-gather of a page-sharded DB build: per page a header
    (width, height, n_keypoints, small_w, small_h), kp_cap keypoints (24 B, slideo_keypoint), kp_cap descriptors (32 B) and
    small_cap bytes of small image.  recs: [(w, h, kp, desc, small)]; returns uint8 [n_slots, stride] (unused slots zero)."""
    stride = PAGE_REC_HEADER * 4 + kp_cap * (24 + 32) + small_cap
    stride = (stride + 15) // 16 * 16
    buf = np.zeros((n_slots, stride), np.uint8)
    for j, (w, h, kp, desc, small) in enumerate(recs):
        n = len(kp)
        assert n <= kp_cap and small.size <= small_cap
        buf[j, : PAGE_REC_HEADER * 4].view(np.int32)[:5] = (w, h, n, small.shape[1], small.shape[0])
        o = PAGE_REC_HEADER * 4
        buf[j, o: o + n * 24] = np.ascontiguousarray(kp).view(np.uint8).reshape(-1)
        o += kp_cap * 24
        buf[j, o: o + n * 32] = np.ascontiguousarray(desc, np.uint8).reshape(-1)
        o += kp_cap * 32
        buf[j, o: o + small.size] = np.ascontiguousarray(small, np.uint8).reshape(-1)
    return buf


def unpack_page_record(row, kp_cap):
    from ._capi import KEYPOINT_DTYPE
    w, h, n, sw, sh = (int(v) for v in row[: PAGE_REC_HEADER * 4].view(np.int32)[:5])
    o = PAGE_REC_HEADER * 4
    kp = row[o: o + n * 24].view(KEYPOINT_DTYPE).copy()
    o += kp_cap * 24
    desc = row[o: o + n * 32].reshape(n, 32).copy()
    o += kp_cap * 32
    small = row[o: o + sw * sh * 3].reshape(sh, sw, 3).copy()
    return w, h, kp, desc, small


def build_page_db_sharded(make_matcher, pages, rank, world, device=None):
    """Page-sharded build of the page DB: rank r analyses pages [lo, hi) of the deck on its GPU, the
    ranks all-gather the per-page records (size, keypoints, descriptors, small image) and every rank assembles the whole
    DB, in page order, from the records (slideo_matcher_add_page_features).  ProcessedImage::compute is independent per page
    (crates/matching-opencv/src/lib.rs:45-47), so the result equals the redundant build bit for bit.
    The exchange is two collectives on plain tensors, no pickling: an all-reduce (MAX) of the two record capacities
    (keypoints per page — ties at the retainBest threshold are kept, so the quota is not a bound — and small-image bytes),
    then ONE all_gather_into_tensor of the fixed-stride records (pack_page_records), on `device` (the GPU under nccl = RCCL).
    make_matcher(): a fresh Matcher with the run's config.  Returns the finalized matcher."""
    import torch
    import torch.distributed as dist
    lo, hi = shard_range(len(pages), rank, world)
    recs = []
    if hi > lo:
        part = make_matcher()
        for i in range(lo, hi, 50):
            part.add_pages(list(pages[i:min(hi, i + 50)]))
        for j in range(hi - lo):
            kp, desc = part.page_features(j)
            recs.append((int(pages[lo + j].shape[1]), int(pages[lo + j].shape[0]), kp, desc, part.page_small(j)))
        part.close()
    caps = torch.tensor([max([len(r[2]) for r in recs] + [1]), max([r[4].size for r in recs] + [1])], dtype=torch.int64)
    if world > 1:
        if device is not None:
            caps = caps.to(device)
        dist.all_reduce(caps, op=dist.ReduceOp.MAX)
    kp_cap, small_cap = int(caps[0]), int(caps[1])
    sizes = [shard_range(len(pages), r, world) for r in range(world)]
    slots = max(b - a for a, b in sizes)
    mine = torch.from_numpy(pack_page_records(recs, kp_cap, small_cap, max(slots, 1)))
    if world > 1:
        if device is not None:
            mine = mine.to(device)
        allb = torch.empty((world * mine.shape[0], mine.shape[1]), dtype=torch.uint8, device=mine.device)
        dist.all_gather_into_tensor(allb, mine)
        allb = allb.cpu().numpy().reshape(world, mine.shape[0], mine.shape[1])
    else:
        allb = mine.numpy()[None]
    m = make_matcher()
    for r, (a, b) in enumerate(sizes):
        for j in range(b - a):
            w, h, kp, desc, small = unpack_page_record(allb[r, j], kp_cap)
            m.add_page_features(w, h, kp, desc, small)
    m.finalize()
    return m
