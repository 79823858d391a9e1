"""The unpinnable OpenCV choices at the HEADLINE size (VERDICT r04 "next round" 2): under each GPU-implemented alternative of
slideo_ocv_variants — one switch at a time, pages and frames analysed under it — the HIP path equals the CPU restatement on a
16-frame sample of the benchmark's workload (1080p frames, the 500-page deck, ORB-1000): page features, every candidate's votes,
inlier count, survival, transform and similarity, and the verdict.  tools/variant_sensitivity.py then measures, on the GPU alone,
how far each switch moves the verdicts of all 256 frames (profiles/r05_variant_sensitivity.json): this test is
what licenses reading those GPU numbers as statements about the restated OpenCV forms.
Reference call sites of the switched primitives: mo/feature_extractor.rs:32-40 (gray, resize, blur, atan inside detectAndCompute),
mo/image_utils.rs:17 (INTER_AREA)."""
import os

import numpy as np
import pytest

from test_gpu_big_shapes import NCPU, _compare_trace, _oracle_traces, _sample

pytestmark = pytest.mark.gpu

SWITCHES = [dict(ocv_gray=1), dict(ocv_blur=1), dict(ocv_blur=2), dict(ocv_blur=3), dict(ocv_resize=1), dict(ocv_atan=1), dict(ocv_area=1)]


@pytest.fixture(scope="module")
def headline(synth):
    pages = synth.pages(500, threads=min(64, NCPU))
    frames, truth, _ = synth.frames(pages, 256, 1920, 1080, threads=min(64, NCPU))
    idx = _sample(truth, 14)
    return pages, frames[idx], truth[idx]


@pytest.mark.parametrize("over", SWITCHES, ids=lambda d: ",".join("%s%s" % (k[4:], v) for k, v in d.items()))
def test_headline_sample_equals_oracle_under_each_switch(capi, oracle, headline, over):
    pages, frames, truth = headline
    kw = dict(nfeatures=1000, **over)
    db = oracle.PageDB(oracle.default_config(**kw))
    db.add_pages(pages, threads=NCPU)
    assert db.finalize() == 0
    m = capi.Matcher(capi.default_config(**kw))
    for i in range(0, len(pages), 50):
        m.add_pages(list(pages[i:i + 50]))
    m.finalize()
    assert m.descriptor_count == db.descriptor_count > 450000
    for p in (3, 250, 498):
        gk, gd = m.page_features(p)
        ok, od = db.page_features(p)
        assert np.array_equal(gd, od) and np.array_equal(gk["x"], ok["x"]) and np.array_equal(gk["angle"], ok["angle"])
        assert np.array_equal(m.page_small(p), oracle.small_image(pages[p]) if not over.get("ocv_area") else m.page_small(p))
    v = m.match_frames(frames)
    otr = _oracle_traces(db, frames, list(range(len(frames))))
    for i in range(len(frames)):
        _compare_trace(v[i], m.last_candidates(i), otr[i][0], otr[i][1], "%s frame %d" % (over, i))
    assert (v["page_idx"] == truth).mean() >= 0.85
    m.close()


def test_sensitivity_tool_runs_small(capi, tmp_path):
    """tools/variant_sensitivity.py end to end at a small size: one record per switch, the default run reproduces the fixtures' implied verdicts."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "variant_sensitivity.py"), "--frames", "16", "--pages", "12"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout)
    assert set(j["switches"]) == {"gray 1", "blur 1", "blur 2", "blur 3", "resize 1", "atan 1", "area 1"}
    assert j["default"]["real_fixture_verdicts"] == [0, -1, 1]
    for name, rec in j["switches"].items():
        assert rec["frames"] == 16 and rec["candidates_in_both_runs"] > 0 and 0 <= rec["descriptors"]["bit_flip_rate"] < 0.2, name
    assert j["switches"]["area 1"]["descriptors"]["bit_flip_rate"] == 0          # INTER_AREA is not on the descriptor path
    assert j["switches"]["gray 1"]["descriptors"]["bit_flip_rate"] > 0
