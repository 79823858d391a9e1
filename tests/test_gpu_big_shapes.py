"""The BASELINE.json configurations at their real shapes, HIP path vs the CPU restatement (VERDICT r01 "next round" item 1).

  headline     1080p x 256 frames in ONE unit vs a 500-page deck (M ~ 517 k train rows): the launch shape bench.py times —
               239 kNN blocks, one train segment; both wave shapes, fused vote filter on and off; decision traces of a
               frame sample against the oracle, properties on all 256 frames
  configs[1]   exactly B = 256 frames vs P = 100 pages, ORB-1000: traces on a 16-frame sample, properties on all
  configs[4]   4K frames, ORB-2000 (the reference's literals), 100 pages of 2001 x 1125: every frame's trace end to end
  configs[3]   the app's flow (changed-frame mask -> match -> sentinel / sort / dedup) over a 1000-page deck, frames sharded
               over TWO ranks that both run the HIP library on one GPU (gloo, 1-frame halo): timeline == one process == truth
The oracle is test infrastructure (oracle/); it only checks here.
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCPU = os.cpu_count() or 1


def _oracle_traces(db, frames, idx):
    with ThreadPoolExecutor(max_workers=min(len(idx), max(1, NCPU // 2))) as ex:      # ctypes calls release the GIL
        return dict(zip(idx, ex.map(lambda i: db.match_frame_trace(frames[i]), idx)))


def _compare_trace(gv, gc, ov, oc, tag):
    assert gv["n_keypoints"] == ov["n_keypoints"], tag
    assert list(gc["page_idx"]) == list(oc["page_idx"]), tag
    assert list(gc["n_votes"]) == list(oc["n_votes"]), tag
    assert list(gc["inliers"]) == list(oc["inliers"]), tag
    assert list(gc["survived"]) == list(oc["survived"]), tag
    for a, b in zip(gc, oc):
        if b["inliers"] > 0:
            scale = np.array([1, 1, 1e3, 1, 1, 1e3, 1e-3, 1e-3, 1])     # 3x3: translations in px, projective terms ~ 1 / px
            assert np.allclose(a["transform"], b["transform"], rtol=1e-5, atol=1e-6 * scale, equal_nan=True), tag
        assert abs(a["similarity"] - b["similarity"]) <= 1e-4, tag
    assert gv["page_idx"] == ov["page_idx"] and gv["inliers"] == ov["inliers"], tag
    assert abs(gv["similarity"] - ov["similarity"]) <= 1e-4, tag


def _one_unit(m, d_frames, n, w, h):
    """all n frames as ONE unit (the shape bench.py submits), traces of every frame kept"""
    import torch
    t = m.submit_dev(d_frames.data_ptr(), n, w, h, stream=torch.cuda.current_stream().cuda_stream)
    v = m.collect(t)
    return v, [m.last_candidates(i) for i in range(n)]


def _sample(truth, k):
    """k spread frame indices that include "no slide" frames when the batch has any"""
    n = len(truth)
    idx = sorted(set(int(x) for x in np.linspace(0, n - 1, k)))
    none = [int(i) for i in np.nonzero(truth < 0)[0][:2]]
    return sorted(set(idx + none))


@pytest.fixture(scope="module")
def deck500(synth):
    return synth.pages(500, threads=min(64, NCPU))


def test_headline_shape_traces_vs_oracle(capi, oracle, synth, deck500):
    import torch
    pages = deck500
    B, fw, fh = 256, 1920, 1080
    frames, truth, _ = synth.frames(pages, B, fw, fh, threads=min(64, NCPU))
    db = oracle.PageDB(oracle.default_config(nfeatures=1000))
    db.add_pages(pages, threads=NCPU)
    assert db.finalize() == 0
    m = capi.Matcher(capi.default_config(nfeatures=1000))
    for i in range(0, 500, 50):
        m.add_pages(list(pages[i:i + 50]))
    m.finalize()
    assert m.descriptor_count == db.descriptor_count > 450000
    for p in (0, 137, 499):
        gk, gd = m.page_features(p)
        ok, od = db.page_features(p)
        assert np.array_equal(gd, od) and np.array_equal(gk["x"], ok["x"]) and np.array_equal(gk["angle"], ok["angle"])
    # the decision trace of EVERY one of the 256 frames against the oracle for the default engine / list mode (the AVX-512 CPU
    # leg does ~4 frames/s per core), a spread sample of 10 for the other three combinations — which must also return identical
    # decisions on all 256 frames
    idx_all, idx_some = list(range(B)), _sample(truth, 10)
    otr = _oracle_traces(db, frames, idx_all)
    d_frames = torch.from_numpy(frames).cuda()
    ref_v = ref_c = None
    for engine in ("mfma2", "mfma4"):
        for exact in (False, True):
            m.set_knn_engine(engine)
            m.set_knn_exact_lists(exact)
            v, cands = _one_unit(m, d_frames, B, fw, fh)
            for i in (idx_all if ref_v is None else idx_some):
                _compare_trace(v[i], cands[i], otr[i][0], otr[i][1], "engine %s exact_lists %s frame %d" % (engine, exact, i))
            if ref_v is None:
                ref_v, ref_c = v, cands
            else:                                     # every engine / list mode: identical decisions on ALL 256 frames
                assert np.array_equal(v, ref_v)
                for a, b in zip(cands, ref_c):
                    assert np.array_equal(a["page_idx"], b["page_idx"]) and np.array_equal(a["n_votes"], b["n_votes"])
                    assert np.array_equal(a["inliers"], b["inliers"]) and np.array_equal(a["similarity"], b["similarity"])
    assert (ref_v["page_idx"] == truth).mean() >= 0.97
    wrong = (ref_v["page_idx"] != truth) & (ref_v["page_idx"] != -1)            # consecutive pages may share a template:
    assert (np.abs(ref_v["page_idx"] - truth)[wrong] <= 2).all()                # a wrong page is a near-duplicate neighbour
    assert 850 < ref_v["n_keypoints"].mean() < 1100
    m.close()


def test_configs1_exact_shape(capi, oracle, synth):
    """BASELINE configs[1]: 1920x1080 frame batch = 256 vs 100 pages, ORB-1000."""
    import torch
    B, P, fw, fh = 256, 100, 1920, 1080
    pages = synth.pages(P, threads=min(64, NCPU))
    frames, truth, _ = synth.frames(pages, B, fw, fh, threads=min(64, NCPU))
    db = oracle.PageDB(oracle.default_config(nfeatures=1000))
    db.add_pages(pages, threads=NCPU)
    assert db.finalize() == 0
    m = capi.Matcher(capi.default_config(nfeatures=1000))
    m.add_pages(list(pages[:50])); m.add_pages(list(pages[50:]))
    m.finalize()
    assert m.descriptor_count == db.descriptor_count
    idx = _sample(truth, 16)
    otr = _oracle_traces(db, frames, idx)
    v, cands = _one_unit(m, torch.from_numpy(frames).cuda(), B, fw, fh)
    for i in idx:
        _compare_trace(v[i], cands[i], otr[i][0], otr[i][1], "frame %d" % i)
    assert (v["page_idx"] == truth).mean() >= 0.97
    # the host-frame entry point cuts the batch into two units: same verdicts
    assert np.array_equal(m.match_frames(frames), v)
    m.close()


def test_configs4_shape_4k_orb2000_end_to_end(capi, oracle, synth):
    """BASELINE configs[4] shape at its deck size: 3840x2160 frames, ORB-2000 = the reference's literals throughout, multi-scale
    pyramid, RANSAC verification, verdicts; 1000 pages (1.8 M train descriptors), 16 frames.  Every frame's decision trace
    against the oracle, and every miss against the synthetic truth explained: the accuracy at this shape (0.875 in bench.py)
    is the reference's absolute acceptance rule `rating > 50` (mo/lib.rs:333) — the true page IS the best-rated candidate, with
    at most 50 inliers — which the oracle applies identically."""
    P, B, fw, fh = 1000, 16, 3840, 2160
    pages = synth.pages(P, threads=min(64, NCPU))
    frames, truth, _ = synth.frames(pages, B, fw, fh, first=40, threads=min(64, NCPU))
    db = oracle.PageDB(oracle.default_config())
    db.add_pages(pages, threads=NCPU)
    assert db.finalize() == 0
    m = capi.Matcher(capi.default_config())
    for i in range(0, P, 50):
        m.add_pages(list(pages[i:i + 50]))
    m.finalize()
    assert m.descriptor_count == db.descriptor_count > 1500000
    otr = _oracle_traces(db, frames, list(range(B)))
    v = m.match_frames(frames)
    misses = 0
    for i in range(B):
        c = m.last_candidates(i)
        _compare_trace(v[i], c, otr[i][0], otr[i][1], "frame %d" % i)
        if v["page_idx"][i] != truth[i]:
            misses += 1
            assert otr[i][0]["page_idx"] == v["page_idx"][i]                      # the oracle misses the same frame the same way
            if truth[i] >= 0 and v["page_idx"][i] == -1:
                best = c[np.argmax(c["inliers"])]
                # (consecutive synthetic pages may share a template, SURVEY 8d: the best-rated page is the true one or its sibling)
                assert abs(int(best["page_idx"]) - int(truth[i])) <= 2 and best["inliers"] <= 50, "a miss that is not the rating > 50 rule"
    assert misses <= B // 4
    assert v["n_keypoints"][truth >= 0].min() > 1500
    m.close()


def _cmp_h_trace(v_i, gc, ov, oc, tag):
    """a homography trace against the oracle's (the matrices of ill-conditioned non-survivors are noise in both: skipped)"""
    assert v_i["n_keypoints"] == ov["n_keypoints"] and list(gc["page_idx"]) == list(oc["page_idx"]), tag
    assert list(gc["n_votes"]) == list(oc["n_votes"]) and list(gc["inliers"]) == list(oc["inliers"]), tag
    assert list(gc["survived"]) == list(oc["survived"]), tag
    for a, b in zip(gc, oc):
        if b["survived"]:
            assert np.allclose(a["transform"], b["transform"], rtol=1e-5, atol=1e-9), tag
        assert abs(a["similarity"] - b["similarity"]) <= 1e-4, tag
    assert v_i["page_idx"] == ov["page_idx"] and v_i["inliers"] == ov["inliers"], tag


@pytest.mark.parametrize("hdlt", [1, 0])
def test_headline_shape_homography_traces_vs_oracle(capi, oracle, synth, deck500, hdlt):
    """The headline shape with the homography verifier (verify_model 1) on perspective frames, 256 frames as ONE unit: 10 240
    candidate slots, so everything size-dependent is on as in bench.py — with ocv.hdlt 1 ransac_h_tail_kernel takes over after the
    default 256 rounds, and refine_h's small candidates get the lane LM (>= 4096 candidates per unit) in both forms.  hdlt 1: the
    traces of ALL 256 frames against the oracle; hdlt 0 (OpenCV's Jacobi form, the default of verify_model 1): the GPU runs the
    same 256-frame unit and 48 frames spread over it are traced (the CPU leg's 9 x 9 eigenproblem per RANSAC sample is what
    bounds the sample).  The whole unit twice: bit-identical."""
    import torch
    pages = deck500
    B, fw, fh = 256, 1920, 1080
    frames, truth, _ = synth.frames_persp(pages, B, fw, fh, persp=0.1, threads=min(64, NCPU))
    kw = dict(nfeatures=1000, verify_model=1, ocv_hdlt=hdlt)
    db = oracle.PageDB(oracle.default_config(**kw))
    db.add_pages(pages, threads=NCPU)
    assert db.finalize() == 0
    m = capi.Matcher(capi.default_config(**kw))
    for i in range(0, 500, 50):
        m.add_pages(list(pages[i:i + 50]))
    m.finalize()
    idx = list(range(B)) if hdlt == 1 else _sample(truth, 48)
    otr = _oracle_traces(db, frames, idx)
    d_frames = torch.from_numpy(frames).cuda()
    v, cands = _one_unit(m, d_frames, B, fw, fh)
    for i in idx:
        _cmp_h_trace(v[i], cands[i], otr[i][0], otr[i][1], "hdlt %d frame %d" % (hdlt, i))
    v2, cands2 = _one_unit(m, d_frames, B, fw, fh)
    assert np.array_equal(v, v2)
    for a, b in zip(cands, cands2):
        assert np.array(a).tobytes() == np.array(b).tobytes()
    assert (v["page_idx"] == truth).mean() >= 0.8          # (the sibling-page effect of the 8-DOF model: DESIGN section 5)
    m.close()


def test_configs4_shape_homography_verification(capi, oracle, synth):
    """configs[4] as BASELINE words it ("RANSAC homography verify") at its deck size: 4K frames generated under a projective map,
    ORB-2000, 1000 pages (1.8 M train descriptors), verify_model 1 in both sample-solver forms; every frame's trace against the
    oracle (csrc/homography.hip.h)."""
    P, B, fw, fh = 1000, 8, 3840, 2160
    pages = synth.pages(P, threads=min(64, NCPU))
    frames, truth, _ = synth.frames_persp(pages, B, fw, fh, persp=0.1, first=7, threads=min(64, NCPU))
    for hdlt in (1, 0):
        kw = dict(verify_model=1, ocv_hdlt=hdlt)
        db = oracle.PageDB(oracle.default_config(**kw))
        db.add_pages(pages, threads=NCPU)
        assert db.finalize() == 0
        m = capi.Matcher(capi.default_config(**kw))
        for i in range(0, P, 50):
            m.add_pages(list(pages[i:i + 50]))
        m.finalize()
        assert m.descriptor_count == db.descriptor_count > 1500000
        v = m.match_frames(frames)
        otr = _oracle_traces(db, frames, list(range(B)))
        for i in range(B):
            _cmp_h_trace(v[i], m.last_candidates(i), otr[i][0], otr[i][1], "hdlt %d frame %d" % (hdlt, i))
        assert (v["page_idx"] == truth).mean() >= 0.5
        m.close()
        del db


@pytest.mark.parametrize("vote", ["tolerance", "ratio"])
def test_configs2_shape_sift_matcher_traces_vs_oracle(capi, oracle, synth, deck500, vote):
    """BASELINE configs[2] at its size as a complete matcher: SIFT-1000 of 1080p frames against the SIFT rows of the 500-page deck
    (0.5 M x 128 B), the squared-L2 search on the int8 matrix cores, then the path's own stages — in both vote rules (the path's
    5 % tolerance vote on 30 rows, bench.py's default; Lowe's ratio test on 2).  The verdicts and candidate traces of 16 frames
    against the CPU restatement in the same mode; until r04 this size was checked by frame 0's SIFT output and 64 neighbour pairs
    inside bench.py only.  The L2 prune bound and the k <= 32 list instance are on by themselves here."""
    from conftest import small_cfg  # noqa: F401
    # (the tolerance vote — bench.py's default — against the whole deck; Lowe's test against its first 150 pages: the CPU leg's SIFT
    # of 500 pages is two minutes per mode)
    pages = deck500 if vote == "tolerance" else deck500[:150]
    B, fw, fh = 16, 1920, 1080
    frames, truth, _ = synth.frames(pages, B, fw, fh, first=3, threads=min(64, NCPU))
    ratio = 0.0 if vote == "tolerance" else 0.75
    sk = dict(nfeatures=1000)
    db = oracle.PageDB(oracle.default_config())
    db.use_sift(oracle.sift_config(**sk), ratio)
    db.add_pages(pages, threads=NCPU)
    assert db.finalize() == 0
    m = capi.Matcher(capi.default_config())
    m.use_sift(capi.sift_config(**sk), ratio)
    for i in range(0, len(pages), 50):
        m.add_pages(list(pages[i:i + 50]))
    m.finalize()
    assert m.descriptor_count == db.descriptor_count > 100000
    for p in (0, 137, len(pages) - 1):
        gk, gd = m.page_features(p)
        ok, od = db.page_features(p)
        assert np.array_equal(gd, od) and np.array_equal(gk, ok.view(gk.dtype)), p
    otr = _oracle_traces(db, frames, list(range(B)))
    v = m.match_frames(frames)
    for i in range(B):
        _compare_trace(v[i], m.last_candidates(i), otr[i][0], otr[i][1], "%s vote, frame %d" % (vote, i))
    acc = (v["page_idx"] == truth).mean()
    assert acc >= (0.9 if vote == "tolerance" else 0.5), acc       # (Lowe's test drops the matches between template-sharing pages: DESIGN section 3)
    m.close()


def test_configs1_shape_lsh_mode_traces_vs_oracle(capi, oracle, synth):
    """The LSH-compatible mode (slideo_config.matcher 1) at configs[1]'s shape — 1080p frames, ORB-1000, 100 pages of 2001 x 1125 —
    where the filtered matrix-core stream runs capacity-sized grids over 93 k rows: traces of 12 frames against the restatement."""
    P, B, fw, fh = 100, 12, 1920, 1080
    pages = synth.pages(P, threads=min(64, NCPU))
    frames, truth, _ = synth.frames(pages, B, fw, fh, first=5, threads=min(64, NCPU))
    kw = dict(nfeatures=1000, matcher=1)
    db = oracle.PageDB(oracle.default_config(**kw))
    db.add_pages(pages, threads=NCPU)
    assert db.finalize() == 0
    m = capi.Matcher(capi.default_config(**kw))
    m.add_pages(list(pages)); m.finalize()
    otr = _oracle_traces(db, frames, list(range(B)))
    v = m.match_frames(frames)
    for i in range(B):
        _compare_trace(v[i], m.last_candidates(i), otr[i][0], otr[i][1], "frame %d" % i)
    assert (v["page_idx"] == truth).mean() >= 0.75
    m.close()


# ---- configs[3] shape: the lecture flow, two ranks on one GPU --------------------------------------------------------------

LECT_PAGES, LECT_SAMPLES = 1000, 96          # 1000-page deck; 8 minutes of lecture sampled every 5 s


def _lecture_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lecture_timeline as LT
    from slideo_amd import synth, distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pages = synth.pages(LECT_PAGES, threads=min(32, NCPU // 2))
    visits = LT.make_visits(LECT_SAMPLES, LECT_PAGES)
    m = LT.build_matcher(pages)                                         # page DB replicated on every rank; BOTH ranks drive the HIP library
    changed, page_of, _, _ = LT.run_shard(m, pages, visits, LECT_SAMPLES, rank, world, batch=32)
    lo, hi = D.shard_range(LECT_SAMPLES, rank, world)
    cap = -(-LECT_SAMPLES // world)
    buf = torch.full((cap, 2), -3, dtype=torch.int32)
    buf[: hi - lo, 0] = torch.from_numpy(changed.astype(np.int32)); buf[: hi - lo, 1] = torch.from_numpy(page_of)
    out = torch.empty((world * cap, 2), dtype=torch.int32)
    dist.all_gather_into_tensor(out, buf)                               # the one collective of the path
    if rank == 0:
        parts = [out[r * cap: r * cap + (D.shard_range(LECT_SAMPLES, r, world)[1] - D.shard_range(LECT_SAMPLES, r, world)[0])] for r in range(world)]
        q.put(torch.cat(parts).numpy().tolist())
    dist.barrier()
    m.close()
    dist.destroy_process_group()


def test_configs3_shape_lecture_two_ranks_one_gpu(capi, synth):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lecture_timeline as LT
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_lecture_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    # meanwhile: the same lecture in ONE process (this one), same GPU
    pages = synth.pages(LECT_PAGES, threads=min(32, NCPU // 2))
    visits = LT.make_visits(LECT_SAMPLES, LECT_PAGES)
    m = LT.build_matcher(pages)
    assert m.page_count == LECT_PAGES and m.descriptor_count > 900000
    changed1, page1, _, _ = LT.run_shard(m, pages, visits, LECT_SAMPLES, 0, 1, batch=32)
    m.close()
    got = np.array(q.get(timeout=900), np.int32)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert np.array_equal(got[:, 0].astype(bool), changed1), "changed flags: two ranks with halo vs one process"
    assert np.array_equal(got[:, 1], page1), "per-sample verdicts: two ranks vs one process"
    imgs = [LT.Page(i + 1) for i in range(LECT_PAGES)]
    tl2 = LT.timeline_from_samples(got[:, 1], LECT_SAMPLES, imgs)
    tl1 = LT.timeline_from_samples(page1, LECT_SAMPLES, imgs)
    tt = LT.truth_timeline(visits, LECT_SAMPLES, imgs)
    k = LT.timeline_key
    assert list(map(k, tl2)) == list(map(k, tl1))
    want, have = set(map(k, tt)), set(map(k, tl2))
    assert len(want - have) <= max(1, len(want) // 20) and len(have - want) <= max(1, len(want) // 20), (sorted(want - have), sorted(have - want))
    assert 0.02 < changed1.mean() < 0.5                                   # the mask removes most sampled frames
    # the oracle's column: changed flags of all 96 samples (video_capture.rs:86-98 restated on the CPU) and the verdict of every
    # changed sample against the CPU restatement over the same 1000-page deck
    import pyoracle
    ocfg = pyoracle.default_config(nfeatures=1000)
    samples = LT.sample_frames(pages, visits, 0, LECT_SAMPLES)
    och = pyoracle.changed_mask(samples, ocfg)[0].astype(bool)
    assert np.array_equal(och, changed1), "changed flags: HIP path vs oracle"
    db = pyoracle.PageDB(ocfg)
    db.add_pages(pages, threads=NCPU)
    assert db.finalize() == 0
    sel = np.flatnonzero(och)
    ov = db.match_frames(samples[sel], threads=min(NCPU, len(sel)))
    assert np.array_equal(ov["page_idx"], page1[sel]), "verdicts of the changed samples: HIP path vs oracle"
    assert (page1[~och] == -2).all()


def test_page_db_from_imported_features_equals_direct_build(capi, synth):
    """The page-sharded build of SURVEY 8e in one process: two 'ranks' analyse half the deck each, the records are imported in
    page order (slideo_matcher_add_page_features) — the assembled matcher equals the one that analysed every page itself."""
    from slideo_amd import distributed as D
    pages = synth.pages(12, 800, 450)
    frames, truth, _ = synth.frames(pages, 10, 640, 360)
    mk = lambda: capi.Matcher(capi.default_config(nfeatures=500, min_rating=12.0))
    direct = mk(); direct.add_pages(list(pages)); direct.finalize()
    recs = []
    for r in range(2):
        lo, hi = D.shard_range(12, r, 2)
        part = mk(); part.add_pages(list(pages[lo:hi]))
        for j in range(hi - lo):
            kp, desc = part.page_features(j)
            recs.append((800, 450, kp, desc, part.page_small(j)))
        part.close()
    imp = mk()
    for w, h, kp, desc, small in recs:
        imp.add_page_features(w, h, kp, desc, small)
    imp.finalize()
    assert imp.page_count == direct.page_count == 12 and imp.descriptor_count == direct.descriptor_count
    for p in (0, 5, 11):
        a, b = imp.page_features(p), direct.page_features(p)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(imp.page_small(p), direct.page_small(p))
    va, vb = imp.match_frames(frames), direct.match_frames(frames)
    assert np.array_equal(va, vb) and (va["page_idx"] == truth).mean() >= 0.8
    # the one-call form (world 1: no collective)
    m1 = D.build_page_db_sharded(mk, pages, 0, 1)
    assert np.array_equal(m1.match_frames(frames), vb)
    with pytest.raises(capi.SlideoError):
        imp2 = mk(); imp2.add_page_features(800, 450, recs[0][2], recs[0][3], recs[0][4][:10])
    for x in (direct, imp, m1):
        x.close()
