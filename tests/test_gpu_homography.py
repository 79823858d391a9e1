"""verify_model 1 (8-DOF homography, BASELINE configs[4] / north_star "RANSAC homography verification"):
the HIP path (csrc/homography.hip.h ransac_h_kernel, reproject_kernel<PERSP>) against the CPU restatement of
cv::findHomography + warpPerspective (oracle find_homography / WarpSampler), through the C ABI.

Bar: candidate pages, vote counts and RANSAC inlier counts equal (the per-sample model is computed in the oracle's exact
operation order), 3x3 transform rel 1e-5 (the refit over the inliers sums in wave order), |d similarity| <= 1e-4, verdicts
equal.  The default model (0, the reference's estimateAffinePartial2D) is covered by every other test file, unchanged."""
import numpy as np
import pytest

from conftest import small_cfg
from test_gpu_parity import _build_both, _compare_traces

pytestmark = pytest.mark.gpu


def _project(H, p):
    q = np.c_[p, np.ones(len(p))] @ H.T
    return q[:, :2] / q[:, 2:]


def test_homography_traces_cfg0(capi, oracle, cfg0_data):
    """BASELINE configs[0] with the homography verifier: every candidate of every frame against the oracle."""
    pages, frames, truth, _ = cfg0_data
    m, db = _build_both(capi, oracle, small_cfg(capi, verify_model=1), small_cfg(oracle, verify_model=1), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v, skip_ill_conditioned=True)
    assert all(g == t or g == -1 for g, t in zip(v["page_idx"], truth))        # (see test_oracle_homography: degenerate strips)
    assert np.array_equal(m.match_frames(frames), v), "deterministic"
    m.close()


@pytest.mark.parametrize("over", [
    dict(ransac_threshold=1.5, ransac_max_iters=300, ransac_confidence=0.9, refine_iters=0),
    dict(knn_k=12, max_candidate_pages=7, max_rated=3, refine_iters=3),
    dict(min_rating=5.0, min_rating_ratio=0.05, min_similarity=0.3, ransac_max_iters=37),
], ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_homography_other_parameters(capi, oracle, cfg0_data, over):
    pages, frames, truth, _ = cfg0_data
    m, db = _build_both(capi, oracle, small_cfg(capi, verify_model=1, **over), small_cfg(oracle, verify_model=1, **over), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v, skip_ill_conditioned=True)
    m.close()


def test_perspective_frames_1080p(capi, oracle, synth):
    """Frames generated under a true projective map (csrc/synth.cpp slideo_synth_frames_persp, keystone ~ 10 %), 2001x1125
    pages, 1080p frames: traces equal the oracle's, the page is found, the generator's homography is recovered, and the
    reference's similarity model keeps far fewer inliers on the same frames."""
    pages = synth.pages(4)
    frames, truth, tH = synth.frames_persp(pages, 6, 1920, 1080, persp=0.2, first=1)
    kw = dict(nfeatures=1000, verify_model=1)
    m, db = _build_both(capi, oracle, capi.default_config(**kw), oracle.default_config(**kw), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v, skip_ill_conditioned=True)
    show = truth >= 0
    # (consecutive synthetic pages may share a template — csrc/synth.cpp, p = 0.3 — and differ by one text row: such a
    # frame can go to the sibling page, in the oracle as here; the traces above are the parity statement)
    right = v["page_idx"] == truth
    assert show.sum() >= 4 and right[show].mean() >= 0.75
    inner = np.array([[300, 200], [1700, 200], [1700, 900], [300, 900], [1000, 560]], np.float64)
    errs = []
    for i in np.flatnonzero(show & right):
        c = m.last_candidates(int(i))
        top = c[np.argmax(c["inliers"])]
        assert top["page_idx"] == truth[i]
        errs.append(np.abs(_project(top["transform"].reshape(3, 3), inner) - _project(tH[i], inner)).max())
    # a page whose keypoints sit in one block of the slide constrains the 8 parameters only there (hundreds of inliers, tens
    # of pixels off elsewhere — in the oracle alike); where they are spread out the generator's homography comes back
    assert min(errs) < 3.0 and np.median(errs) < 60.0, errs
    m0 = capi.Matcher(capi.default_config(nfeatures=1000))
    m0.add_pages(list(pages)); m0.finalize()
    v0 = m0.match_frames(frames)
    assert v["inliers"][show & right].sum() > 1.3 * v0["inliers"][show & right].sum()  # (a 10 % keystone leaves the 4-DOF model a fraction of the votes)
    m0.close(); m.close()


def test_homography_larger_deck_and_units(capi, oracle, synth):
    """48 pages / 24 frames (candidates with 5..200 votes: both LDS instances, duplicate-heavy sample schedules of tiny
    candidates, "no slide" frames) and the same batch cut into pipelined units."""
    pages = synth.pages(48, 800, 450)
    frames, truth, _ = synth.frames(pages, 24, 640, 360)
    m, db = _build_both(capi, oracle, small_cfg(capi, verify_model=1), small_cfg(oracle, verify_model=1), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v, skip_ill_conditioned=True)
    big = np.concatenate([frames] * 6)                      # 144 frames: two halves in flight
    vb = m.match_frames(big)
    assert np.array_equal(vb, np.concatenate([v] * 6))
    m.close()


def test_homography_rng_stream_grows_on_demand(capi, oracle, cfg0_data, monkeypatch):
    """A stream far too short for the 4-point schedules (rejected subsets consume draws too): grown and re-run."""
    pages, frames, truth, _ = cfg0_data
    monkeypatch.setenv("SLIDEO_RNG_STREAM_LEN", "2048")
    m, db = _build_both(capi, oracle, small_cfg(capi, verify_model=1), small_cfg(oracle, verify_model=1), pages)
    monkeypatch.delenv("SLIDEO_RNG_STREAM_LEN")
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v, skip_ill_conditioned=True)
    m.close()


def test_homography_config_validation(capi):
    with pytest.raises(capi.SlideoError) as e:
        capi.Matcher(capi.default_config(verify_model=2))
    assert e.value.code == 5
    with pytest.raises(capi.SlideoError) as e:
        capi.Matcher(capi.default_config(verify_model=1, ocv_hdlt=3))
    assert e.value.code == 5


@pytest.mark.parametrize("hdlt", [1, 2])
def test_direct_sample_solver_hdlt1(capi, oracle, synth, cfg0_data, hdlt):
    """ocv.hdlt 1 / 2 (8x8 elimination / closed form per minimal sample instead of the Jacobi sweep): GPU == the oracle's same
    form, and the same verdicts and inlier counts as the default form 0 on these inputs."""
    pages, frames, truth, _ = cfg0_data
    kw = dict(verify_model=1, ocv_hdlt=hdlt)
    m, db = _build_both(capi, oracle, small_cfg(capi, **kw), small_cfg(oracle, **kw), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v, skip_ill_conditioned=True)
    m.close()
    pages = synth.pages(48, 800, 450)
    frames, truth, _ = synth.frames(pages, 24, 640, 360)
    m, db = _build_both(capi, oracle, small_cfg(capi, **kw), small_cfg(oracle, **kw), pages)
    v1 = m.match_frames(frames)
    _compare_traces(m, db, frames, v1, skip_ill_conditioned=True)
    m.close()
    m0 = capi.Matcher(small_cfg(capi, verify_model=1))
    m0.add_pages(list(pages)); m0.finalize()
    v0 = m0.match_frames(frames)
    assert np.array_equal(v0["page_idx"], v1["page_idx"]) and np.array_equal(v0["inliers"], v1["inliers"])
    m0.close()


@pytest.mark.parametrize("hdlt", [1, 2])
def test_tail_kernel_equals_single_wave(capi, oracle, synth, cfg0_data, hdlt, monkeypatch):
    """ransac_h_tail_kernel (8 waves on the sample schedule of a candidate that ransac_h_kernel gave up on after
    SLIDEO_RH_TAIL_ROUNDS sampling rounds): with the cap at 1 every sampled candidate goes through it, with 3 a mixture —
    verdicts and full traces equal the single-wave kernel's (cap off) and the oracle's."""
    pages = synth.pages(48, 800, 450)
    frames, truth, _ = synth.frames(pages, 24, 640, 360)
    kw = dict(verify_model=1, ocv_hdlt=hdlt)
    runs = {}
    for cap in ("0", "1", "3"):
        monkeypatch.setenv("SLIDEO_RH_TAIL_ROUNDS", cap)
        m, db = _build_both(capi, oracle, small_cfg(capi, **kw), small_cfg(oracle, **kw), pages)
        v = m.match_frames(frames)
        _compare_traces(m, db, frames, v, skip_ill_conditioned=True)
        runs[cap] = (v, [np.array(m.last_candidates(i)) for i in range(len(frames))])
        m.close()
    for cap in ("1", "3"):
        assert np.array_equal(runs[cap][0], runs["0"][0]), cap
        for ca, cb in zip(runs[cap][1], runs["0"][1]):
            assert ca.tobytes() == cb.tobytes(), "candidates (inliers, models, ratings) bit-identical to the single-wave kernel's"
    # the perspective frames (more votes per candidate: the LDS-points instance too)
    pages = synth.pages(4)
    frames, truth, tH = synth.frames_persp(pages, 6, 1920, 1080, persp=0.2, first=1)
    kw = dict(nfeatures=1000, verify_model=1, ocv_hdlt=hdlt)
    monkeypatch.setenv("SLIDEO_RH_TAIL_ROUNDS", "1")
    m, db = _build_both(capi, oracle, capi.default_config(**kw), oracle.default_config(**kw), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v, skip_ill_conditioned=True)
    m.close()


def test_tail_kernel_candidates_beyond_the_lds_points(capi, oracle, synth, monkeypatch):
    """ORB-4000 on 1080p frames: the true page collects more votes than RANSAC_LDS_PTS (1024), so its point pairs live in
    global memory (gpts) — in ransac_h_kernel's large instance and, with the hand-over cap at 1, in ransac_h_tail_kernel.
    Traces equal the oracle's and the cap-off run's bit for bit."""
    pages = synth.pages(4)
    frames, truth, _ = synth.frames(pages, 3, 1920, 1080, first=1)
    kw = dict(nfeatures=4000, verify_model=1, ocv_hdlt=1)
    runs = {}
    for cap in ("0", "1"):
        monkeypatch.setenv("SLIDEO_RH_TAIL_ROUNDS", cap)
        m, db = _build_both(capi, oracle, capi.default_config(**kw), oracle.default_config(**kw), pages)
        v = m.match_frames(frames)
        _compare_traces(m, db, frames, v, skip_ill_conditioned=True)
        runs[cap] = (v, [np.array(m.last_candidates(i)) for i in range(len(frames))])
        m.close()
    assert max(int(c["n_votes"].max()) for c in runs["1"][1] if len(c)) > 1024, "the case this test is for"
    assert np.array_equal(runs["0"][0], runs["1"][0])
    for ca, cb in zip(runs["0"][1], runs["1"][1]):
        assert ca.tobytes() == cb.tobytes()


def test_lane_lm_equals_wave_lm(capi, oracle, synth, monkeypatch):
    """refine_h_eigen_kernel's lane-per-candidate LM (candidates with <= 48 votes; SLIDEO_REFINE_LANE_LM=1) against the
    wave-per-candidate LM of refine_h_kernel<1> (=0): verdicts and candidate records bit-identical, both equal to the oracle."""
    pages = synth.pages(48, 800, 450)
    frames, truth, _ = synth.frames(pages, 24, 640, 360)
    kw = dict(verify_model=1, ocv_hdlt=1)
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SLIDEO_REFINE_LANE_LM", mode)
        m, db = _build_both(capi, oracle, small_cfg(capi, **kw), small_cfg(oracle, **kw), pages)
        v = m.match_frames(frames)
        _compare_traces(m, db, frames, v, skip_ill_conditioned=True)
        runs[mode] = (v, [np.array(m.last_candidates(i)) for i in range(len(frames))])
        m.close()
    assert np.array_equal(runs["0"][0], runs["1"][0])
    for ca, cb in zip(runs["0"][1], runs["1"][1]):
        assert ca.tobytes() == cb.tobytes()


def test_sample_solver_forms_agree_end_to_end(capi, synth):
    """hdlt 1 / 2 against cv::findHomography's own form (hdlt 0), end to end on 1080p perspective frames against a 60-page deck: the
    models of a sample agree to f64 round-off, so candidates, inlier counts and verdicts agree except where round-off flips an
    inlier at the 3 px threshold (tools/hdlt_agreement.py measures the headline shape: profiles/r04_hdlt_agreement.json).
    This is what licenses bench.py --workload cfg4 to default to the cheap form."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hdlt_agreement as HA
    pages = synth.pages(60)
    frames, truth, _ = synth.frames_persp(pages, 24, 1920, 1080, persp=0.1, seed=3)
    res = {h: HA.run(pages, frames, h, 1000) for h in (0, 1, 2)}
    for h in (1, 2):
        c = HA.compare(res[h], res[0])
        assert c["verdict_page_agreement"] >= 0.95 and c["candidate_inlier_count_agreement"] >= 0.95, (h, c)
        assert c["max_similarity_difference"] <= 0.05, (h, c)


@pytest.mark.parametrize("model", [0, 1])
def test_verdict_rule_1_equals_oracle(capi, oracle, synth, model):
    """slideo_config.verdict_rule 1 (opt-in departure from mo/lib.rs:370-389: the survivors keep their rating order and the
    re-projection similarity only accepts): verdict_kernel == the restatement, both verify models, thresholds loosened so that
    frames have several accepted survivors (template-sharing sibling pages)."""
    pages = synth.pages(24, 800, 450)
    frames, truth, _ = synth.frames(pages, 24, 640, 360)
    min_sim = 0.05
    kw = dict(nfeatures=500, min_rating=6.0, min_rating_ratio=0.02, min_similarity=min_sim, verify_model=model, verdict_rule=1)
    m, db = _build_both(capi, oracle, capi.default_config(**kw), oracle.default_config(**kw), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v, skip_ill_conditioned=True)
    several = 0
    for i in range(len(frames)):
        c = m.last_candidates(i)
        ok = c[(c["survived"] == 1) & (c["similarity"] > min_sim)]       # rule 1: the similarity only ACCEPTS (the configured bound) ...
        several += len(ok) > 1
        if len(ok):
            assert v["page_idx"][i] == ok[np.argmax(ok["inliers"])]["page_idx"]      # ... and the first in rating (= inlier) order wins
    assert several >= 1, "no frame exercised the rule"
    m.close()
    with pytest.raises(capi.SlideoError):
        capi.Matcher(capi.default_config(verdict_rule=2))
