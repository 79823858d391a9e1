"""slideo_matcher_use_sift — north-star / BASELINE configs[2] as a complete matcher: SIFT features, squared-L2 2-NN on the int8
matrix cores, Lowe's ratio test, then the path's own vote / RANSAC / re-projection stages.  Parity target: the CPU restatement in
the same mode (so_pagedb_use_sift).  Bar: page features bit-exact, votes / inliers / verdicts equal, |d similarity| <= 1e-4."""
import numpy as np
import pytest

from conftest import small_cfg
from test_gpu_parity import _compare_traces

pytestmark = pytest.mark.gpu


def _build(capi, oracle, pages, sift_kw, ratio, **over):
    m = capi.Matcher(small_cfg(capi, **over))
    m.use_sift(capi.sift_config(**sift_kw), ratio)
    m.add_pages(list(pages))
    m.finalize()
    db = oracle.PageDB(small_cfg(oracle, **over))
    db.use_sift(oracle.sift_config(**sift_kw), ratio)
    for p in pages:
        db.add_page(p)
    assert db.finalize() == 0
    return m, db


def test_sift_matcher_end_to_end_cfg0(capi, oracle, cfg0_data):
    pages, frames, truth, _ = cfg0_data
    m, db = _build(capi, oracle, pages, dict(nfeatures=400), 0.75)
    assert m.descriptor_count == db.descriptor_count > 0
    for p in range(len(pages)):
        gk, gd = m.page_features(p)
        ok, od = db.page_features(p)
        assert gd.shape[1] == 128 and np.array_equal(gd, od) and np.array_equal(gk, ok.view(gk.dtype))
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v)
    assert list(v["page_idx"]) == list(truth), "page assignment vs synthetic ground truth"
    # the streaming entry points and the host path give the same verdicts
    import torch
    d = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    t1 = m.submit_dev(d.data_ptr(), len(frames), frames.shape[2], frames.shape[1])
    t2 = m.submit_dev(d.data_ptr(), len(frames), frames.shape[2], frames.shape[1])
    assert np.array_equal(m.collect(t1), v) and np.array_equal(m.collect(t2), v)
    m.close()


@pytest.mark.parametrize("mode", ["similarity", "homography", "all_keypoints", "tolerance_vote"])
def test_sift_matcher_modes(capi, oracle, synth, mode):
    pages = synth.pages(6, 800, 450, seed=11)
    over, sk, ratio = {}, dict(nfeatures=300), 0.8
    if mode == "homography":
        over = dict(verify_model=1, ocv_hdlt=1)
        frames, truth, _ = synth.frames_persp(pages, 5, 640, 360, persp=0.1, seed=5)
    else:
        frames, truth, _ = synth.frames(pages, 5, 640, 360, seed=5)
    if mode == "all_keypoints":
        sk, ratio = dict(nfeatures=0, contrast_threshold=0.06), 0.7
    if mode == "tolerance_vote":                                   # ratio 0: the path's own vote rule on the L2 distances, knn_k rows
        ratio = 0.0
        pages = np.concatenate([pages, pages[:2]])                 # twin pages: Lowe's test would drop their matches
        over = dict(knn_k=30, vote_tolerance=1.05)
    m, db = _build(capi, oracle, pages, sk, ratio, **over)
    assert m.descriptor_count == db.descriptor_count > 0
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v, skip_ill_conditioned=mode == "homography")
    m.close()


def test_sift_matcher_state_and_limits(capi, synth):
    pages = synth.pages(2, 640, 360, seed=3)
    m = capi.Matcher(small_cfg(capi))
    m.add_pages(list(pages[:1]))
    with pytest.raises(capi.SlideoError) as e:
        m.use_sift(capi.sift_config(), 0.75)                       # after the first page
    assert e.value.code == 4
    m.close()
    m = capi.Matcher(small_cfg(capi))
    with pytest.raises(capi.SlideoError):
        m.use_sift(capi.sift_config(), 1.5)
    m.use_sift(capi.sift_config(nfeatures=200), 0.75)
    m.add_pages(list(pages))
    m.finalize()
    assert m.descriptor_count > 0
    frames, _, _ = synth.frames(pages, 4, 640, 360, seed=9)
    v = m.match_frames(frames)
    assert len(v) == 4
    # the changed-mask call's upload matched in place (kept frames) gives the same verdicts
    changed, _, _ = m.changed_mask(frames, None)
    sel = np.array([3, 0, 2], np.int32)
    assert np.array_equal(m.match_kept_frames(sel), v[sel])
    with pytest.raises(capi.SlideoError) as e:                     # the 32-byte page-feature import is ORB's
        m2 = capi.Matcher(small_cfg(capi)); m2.use_sift(capi.sift_config(), 0.75)
        m2.add_page_features(640, 360, np.zeros(0, capi.KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8), np.zeros((259, 461, 3), np.uint8))
    assert e.value.code == 5
    # a frame side above 4095 (the doubled image's coordinates travel in 13 bits) is refused, not matched wrongly;
    # the ORB path takes the same frame
    wide = np.zeros((40, 4096, 3), np.uint8)
    with pytest.raises(capi.SlideoError) as e:
        m.match_frames(wide[None])
    assert e.value.code == 5 and "4095" in str(e.value)
    m.close()
