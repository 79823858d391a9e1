"""cv::SIFT::detectAndCompute restated on the CPU (oracle/sift_oracle.h) — BASELINE configs[2]'s extractor, which the reference does
not have: definitions pinned against numpy / hand-computable cases, and its usefulness against the synthetic truth."""
import numpy as np


def test_sift_pyramid_shapes_and_first_layers(oracle, synth):
    pages = synth.pages(1, 800, 450)
    img = pages[0]
    g0 = oracle.sift_layer(img, 0, 0)
    assert g0.shape == (900, 1600)                                          # doubled first octave (firstOctave = -1)
    g13 = oracle.sift_layer(img, 1, 0)
    assert g13.shape == (450, 800) and np.array_equal(g13, oracle.sift_layer(img, 0, 3)[::2, ::2])     # next octave = every second pixel of layer 3
    d0 = oracle.sift_layer(img, 0, 0, dog=True)
    assert np.array_equal(d0, oracle.sift_layer(img, 0, 1) - g0)
    # the blurred base is a smoothed 2x bilinear upsample of the gray image: its mean is the gray image's (to f32 noise)
    gray = oracle.gray(img).astype(np.float64)
    assert abs(g0.mean() - gray.mean()) < 0.05
    # a constant image stays constant through every layer and has no extremum
    flat = np.full((120, 160, 3), 77, np.uint8)
    assert np.allclose(oracle.sift_layer(flat, 0, 5), oracle.gray(flat)[0, 0], atol=1e-3)
    kp, desc, st = oracle.sift(flat)
    assert len(kp) == 0 and st["extrema"] == 0


def test_sift_blob_is_found_at_its_place_and_scale(oracle):
    """A Gaussian blob of sigma s0: one dominant extremum at the blob's centre whose keypoint size tracks 2 * sqrt(2) * s0."""
    h, w = 240, 320
    yy, xx = np.mgrid[0:h, 0:w]
    for s0, (cx, cy) in ((6.0, (150, 110)), (11.0, (170, 125))):
        blob = 40 + 180 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s0 * s0))
        img = np.repeat(np.clip(np.rint(blob), 0, 255).astype(np.uint8)[:, :, None], 3, 2)
        kp, desc, st = oracle.sift(img)
        assert len(kp) >= 1
        best = kp[np.argmax(kp["response"])]
        assert abs(best["x"] - cx) < 1.5 and abs(best["y"] - cy) < 1.5
        assert 0.6 < best["size"] / (2 * np.sqrt(2) * s0) < 1.6
        assert desc.shape == (len(kp), 128) and desc.max() <= 255 and np.all(np.abs(np.linalg.norm(desc.astype(np.float64), axis=1) - 512) < 40)


def test_sift_retain_best_and_canonical_order(oracle, synth):
    pages = synth.pages(1, 800, 450)
    kp, desc, _ = oracle.sift(pages[0])
    assert len(kp) > 400
    n = 200
    kp2, desc2, _ = oracle.sift(pages[0], oracle.sift_config(nfeatures=n))
    thr = np.sort(kp["response"])[::-1][n - 1]
    keep = kp["response"] >= thr                                             # ties at the n-th response are kept
    assert keep.sum() >= n and np.array_equal(kp2, kp[keep]) and np.array_equal(desc2, desc[keep])
    # canonical order: (octave, layer) blocks ascending; octave -1 (the doubled image) first, stored as 255
    o = kp["octave"] & 255
    o = np.where(o >= 128, o.astype(np.int32) - 256, o)
    lay = (kp["octave"] >> 8) & 255
    assert np.all(np.diff(o * 8 + lay) >= 0) and o.min() == -1 and set(lay.tolist()) <= {1, 2, 3}
    assert np.all((kp["angle"] >= 0) & (kp["angle"] < 360))


def test_sift_matches_a_transformed_view(oracle, synth):
    """Frame = page under a similarity transform + noise: ratio-test matches of the descriptors are geometrically consistent."""
    pages = synth.pages(2, 800, 450)
    frames, truth, tm = synth.frames(pages, 2, 640, 360)
    i = int(np.flatnonzero(truth >= 0)[0]); pg = int(truth[i])
    kf, df, _ = oracle.sift(frames[i])
    kp, dp, _ = oracle.sift(pages[pg])
    idx, dist = oracle.knn_l2_u8(df, dp, 2)
    good = np.sqrt(dist[:, 0].astype(np.float64)) < 0.75 * np.sqrt(dist[:, 1].astype(np.float64))
    assert good.sum() > 100
    M = np.vstack([tm[i], [0, 0, 1]])
    proj = np.c_[kp["x"][idx[good, 0]], kp["y"][idx[good, 0]], np.ones(good.sum())] @ M.T
    err = np.hypot(proj[:, 0] - kf["x"][good], proj[:, 1] - kf["y"][good])
    assert (err < 3).mean() > 0.95
    # against the WRONG page the ratio test leaves almost nothing
    ko, do, _ = oracle.sift(pages[1 - pg])
    idx2, dist2 = oracle.knn_l2_u8(df, do, 2)
    good2 = np.sqrt(dist2[:, 0].astype(np.float64)) < 0.75 * np.sqrt(dist2[:, 1].astype(np.float64))
    assert good2.sum() < good.sum() / 3


def test_sift_matcher_mode_end_to_end_cfg0(oracle, cfg0_data):
    """so_pagedb_use_sift: SIFT features + squared-L2 2-NN + Lowe's ratio test in front of the path's own stages — the frames of
    configs[0] are assigned to their pages, the 'no slide' ones to none, and the ratio decides who votes."""
    from conftest import small_cfg
    pages, frames, truth, _ = cfg0_data
    db = oracle.PageDB(small_cfg(oracle))
    db.use_sift(oracle.sift_config(nfeatures=400), 0.75)
    for p in pages:
        db.add_page(p)
    assert db.finalize() == 0 and db.descriptor_count > 0
    kp, desc = db.page_features(0)
    assert desc.shape[1] == 128 and len(kp) == len(desc) > 0
    v = db.match_frames(frames)
    assert list(v["page_idx"]) == list(truth)
    # a stricter ratio lets fewer queries vote
    loose, strict = [], []
    for r, acc in ((0.9, loose), (0.5, strict)):
        d2 = oracle.PageDB(small_cfg(oracle))
        d2.use_sift(oracle.sift_config(nfeatures=400), r)
        for p in pages:
            d2.add_page(p)
        d2.finalize()
        _, cands = d2.match_frame_trace(frames[1])
        acc.append(int(cands["n_votes"].sum()))
    assert strict[0] < loose[0]
