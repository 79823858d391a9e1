"""The reference's only real fixtures (data/matchings/test1, copied to tests/golden/) and the
committed expected.json (tests/golden/make_golden.py)."""
import hashlib
import json
import os

import ctypes as C

import numpy as np
import pytest
from PIL import Image

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXP = json.load(open(os.path.join(HERE, "expected.json")))
PAGES = ["1-slide.png", "3-slide.png"]


def load(name):
    return np.ascontiguousarray(np.array(Image.open(os.path.join(HERE, name)).convert("RGB"))[:, :, ::-1])


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_fixture_images_are_the_reference_files():
    for name, e in list(EXP["pages"].items()) + list(EXP["frames"].items()):
        assert sha(load(name)) == e["image_sha256"], name


def test_oracle_reproduces_implied_verdicts_and_pins(oracle):
    cfg = oracle.default_config()
    db = oracle.PageDB(cfg)
    for p in PAGES:
        img = load(p)
        kp, desc = oracle.orb(img, cfg)
        e = EXP["pages"][p]
        assert len(kp) == e["n_keypoints"] and sha(desc) == e["desc_sha256"]
        assert sha(oracle.small_image(img)) == e["small_sha256"]
        db.add_page(img)
    assert db.finalize() == 0 and db.descriptor_count == EXP["descriptor_count"]
    for f, e in EXP["frames"].items():
        v, cands = db.match_frame_trace(load(f))
        # the only reference-side pin: the verdict implied by the fixture's file name
        assert int(v["page_idx"]) == e["implied_page"], f
        assert int(v["page_idx"]) == e["verdict"]["page_idx"] and int(v["inliers"]) == e["verdict"]["inliers"]
        assert abs(float(v["similarity"]) - e["verdict"]["similarity"]) < 1e-6
        assert [int(c["n_votes"]) for c in cands] == [c["n_votes"] for c in e["candidates"]]
        assert [int(c["inliers"]) for c in cands] == [c["inliers"] for c in e["candidates"]]


def test_synthetic_cfg0_pin(oracle, cfg0_data):
    pages, frames, truth, _ = cfg0_data
    e = EXP["synthetic_cfg0"]
    assert sha(pages) == e["pages_sha256"] and sha(frames) == e["frames_sha256"], "synthetic generator drifted"
    assert truth.tolist() == e["truth"]
    c0 = oracle.default_config(nfeatures=500, min_rating=12.0)
    db = oracle.PageDB(c0)
    db.add_pages(pages, threads=4)
    assert db.finalize() == 0
    assert db.descriptor_count == e["descriptor_count"] and sha(db.train()) == e["train_sha256"]
    v = db.match_frames(frames, threads=4)
    assert v["page_idx"].tolist() == e["page_idx"] == e["truth"]
    assert v["inliers"].tolist() == e["inliers"] and v["n_keypoints"].tolist() == e["n_keypoints"]


def test_ratio_test_option_on_cfg0(oracle, cfg0_data):
    """Extension (include/slideo_amd.h `ratio_test`): the ratio rule gives at most one vote per query, never more
    votes than the reference's tolerance rule, and still finds the right pages on the synthetic set."""
    pages, frames, truth, _ = cfg0_data
    res = {}
    for r in (0.0, 0.8):
        db = oracle.PageDB(oracle.default_config(nfeatures=500, min_rating=12.0, ratio_test=r))
        db.add_pages(pages, threads=4)
        assert db.finalize() == 0
        res[r] = [db.match_frame_trace(f) for f in frames]
    for (v0, c0), (v1, c1) in zip(res[0.0], res[0.8]):
        assert int(c1["n_votes"].sum()) <= int(c0["n_votes"].sum())
        assert int(c1["n_votes"].sum()) <= int(v1["n_keypoints"])
    got = np.array([int(v["page_idx"]) for v, _ in res[0.8]])
    assert ((got == truth) | (got == -1)).all() and (got == truth).mean() >= 0.75     # fewer votes: a miss is "none"
    assert oracle.lib().so_config_supported(C.byref(oracle.default_config(ratio_test=0.8, knn_k=1))) == 0


@pytest.mark.gpu
def test_gpu_matches_golden_fixtures(capi):
    """HIP path on the reference's real frames/slides: equals the committed expectations."""
    m = capi.Matcher(capi.default_config())
    m.add_pages([load(p) for p in PAGES])
    m.finalize()
    assert m.descriptor_count == EXP["descriptor_count"]
    for i, p in enumerate(PAGES):
        kp, desc = m.page_features(i)
        assert len(kp) == EXP["pages"][p]["n_keypoints"] and sha(desc) == EXP["pages"][p]["desc_sha256"]
        assert sha(np.stack([kp["x"], kp["y"]], 1)) == EXP["pages"][p]["kp_xy_sha256"]
        assert sha(m.small_image(load(p))) == EXP["pages"][p]["small_sha256"]
    names = list(EXP["frames"])
    v = m.match_frames(np.stack([load(f) for f in names]))
    for i, f in enumerate(names):
        e = EXP["frames"][f]
        kp, desc = m.orb(load(f))
        assert len(kp) == e["n_keypoints"] and sha(desc) == e["desc_sha256"]
        assert int(v[i]["page_idx"]) == e["implied_page"] == e["verdict"]["page_idx"]
        assert int(v[i]["inliers"]) == e["verdict"]["inliers"]
        assert abs(float(v[i]["similarity"]) - e["verdict"]["similarity"]) <= 1e-4
        c = m.last_candidates(i)
        assert [int(x) for x in c["n_votes"]] == [x["n_votes"] for x in e["candidates"]]
        assert [int(x) for x in c["inliers"]] == [x["inliers"] for x in e["candidates"]]
    m.close()
