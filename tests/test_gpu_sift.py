"""SIFT on the GPU (csrc/sift.hip.h) against the CPU restatement of cv::SIFT::detectAndCompute (oracle/sift_oracle.h), through
the C ABI.  Bar: pyramid layers, keypoints (every field) and the 128 descriptor bytes BIT-EXACT — every float operation is in the
oracle's order and the histograms are fixed-point sums."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m(capi):
    mm = capi.Matcher(capi.default_config())
    yield mm
    mm.close()


def _cmp(capi, oracle, m, img, **sc):
    gk, gd = m.sift(img, capi.sift_config(**sc))
    ok, od, st = oracle.sift(img, oracle.sift_config(**sc))
    assert len(gk) == len(ok), (len(gk), len(ok), st)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(gk[f], ok[f]), f
    assert np.array_equal(gd, od)
    return len(gk)


def test_sift_pyramid_bit_exact(capi, oracle, m, cfg0_data):
    pages, frames, _, _ = cfg0_data
    for img in (frames[1], pages[0][:333, :517].copy()):
        for (o, l, dog) in ((0, 0, False), (0, 1, False), (0, 5, False), (1, 0, False), (2, 3, False), (0, 0, True), (1, 4, True), (4, 2, True), (6, 1, False)):
            a, b = m.sift_layer(img, o, l, dog), oracle.sift_layer(img, o, l, dog)
            assert a.shape == b.shape and np.array_equal(a, b), (o, l, dog)


def test_sift_bit_exact_cfg0_images(capi, oracle, m, cfg0_data):
    pages, frames, truth, _ = cfg0_data
    n = 0
    for img in list(frames[:4]) + [pages[0], pages[3]]:
        n += _cmp(capi, oracle, m, img)
    assert n > 2000


def test_sift_nfeatures_and_other_parameters(capi, oracle, m, cfg0_data):
    pages, frames, _, _ = cfg0_data
    assert _cmp(capi, oracle, m, pages[1], nfeatures=150) >= 150
    _cmp(capi, oracle, m, pages[1], nfeatures=1)
    _cmp(capi, oracle, m, frames[2], contrast_threshold=0.08, edge_threshold=5.0)
    _cmp(capi, oracle, m, frames[2], sigma=1.2)


def test_sift_edge_cases(capi, oracle, m):
    rng = np.random.default_rng(5)
    assert _cmp(capi, oracle, m, np.full((90, 140, 3), 200, np.uint8)) == 0               # flat: nothing
    _cmp(capi, oracle, m, rng.integers(0, 256, (97, 131, 3), dtype=np.uint8))               # noise, odd sizes
    _cmp(capi, oracle, m, rng.integers(0, 256, (24, 40, 3), dtype=np.uint8))                # smaller than most octaves' borders
    img = np.zeros((200, 300, 3), np.uint8); img[60:140, 100:220] = 255                     # one hard rectangle: many exact ties in the DoG
    _cmp(capi, oracle, m, img)
    with pytest.raises(capi.SlideoError) as e:
        m.sift(img, capi.sift_config(n_octave_layers=4))
    assert e.value.code == 5


def test_sift_1080p_and_batch_device_path(capi, oracle, m, synth):
    import torch
    pages = synth.pages(2)
    frames, truth, _ = synth.frames(pages, 3, 1920, 1080, first=2)
    sc = dict(nfeatures=1000)
    ok, od, _ = oracle.sift(frames[0], oracle.sift_config(**sc))
    gk, gd = m.sift(frames[0], capi.sift_config(**sc))
    assert len(gk) == len(ok) >= 1000 and np.array_equal(gd, od) and np.array_equal(gk["x"], ok["x"]) and np.array_equal(gk["angle"], ok["angle"])
    # the batch entry point: device frames in, packed device arrays out, frame after frame
    d = torch.from_numpy(frames).cuda()
    cap = 3 * 1400
    kp = torch.zeros((cap, 6), dtype=torch.int32, device="cuda")
    desc = torch.zeros((cap, 128), dtype=torch.uint8, device="cuda")
    qofs, ms = m.sift_frames_dev(d.data_ptr(), 3, 1920, 1080, kp.data_ptr(), desc.data_ptr(), cap, capi.sift_config(**sc))
    assert qofs[0] == 0 and qofs[1] == len(gk) and ms > 0
    hd = desc.cpu().numpy(); hk = kp.cpu().numpy().view(capi.KEYPOINT_DTYPE).reshape(-1)
    assert np.array_equal(hd[: qofs[1]], gd) and np.array_equal(hk[: qofs[1]], gk)
    k2, d2 = m.sift(frames[2], capi.sift_config(**sc))
    assert np.array_equal(hd[qofs[2]: qofs[3]], d2) and np.array_equal(hk[qofs[2]: qofs[3]], k2)
    with pytest.raises(capi.SlideoError) as e:
        m.sift_frames_dev(d.data_ptr(), 3, 1920, 1080, kp.data_ptr(), desc.data_ptr(), 100, capi.sift_config(**sc))
    assert e.value.code == 7


def test_sift_sub_batches_equal_single_frames(capi, oracle, synth, monkeypatch):
    """More frames than one pass holds: the batch entry point walks over sub-batches (24 GB of pyramids each; here the budget is
    squeezed to two frames per pass) — every frame's rows must be what the single-image entry point returns."""
    import torch
    monkeypatch.setenv("SLIDEO_SIFT_WS_MB", "24")                      # 640 x 360: ~11 MB per frame -> 2 frames per pass
    pages = synth.pages(2)
    frames, truth, _ = synth.frames(pages, 5, 640, 360, first=7)
    m = capi.Matcher(capi.default_config())
    sc = capi.sift_config(nfeatures=300)
    d = torch.from_numpy(frames).cuda()
    cap = 5 * 600
    kp = torch.zeros((cap, 6), dtype=torch.int32, device="cuda")
    desc = torch.zeros((cap, 128), dtype=torch.uint8, device="cuda")
    qofs, ms = m.sift_frames_dev(d.data_ptr(), 5, 640, 360, kp.data_ptr(), desc.data_ptr(), cap, sc)
    hd = desc.cpu().numpy(); hk = kp.cpu().numpy().view(capi.KEYPOINT_DTYPE).reshape(-1)
    assert len(qofs) == 6 and qofs[0] == 0
    for i in range(5):
        k1, d1 = m.sift(frames[i], sc)
        assert qofs[i + 1] - qofs[i] == len(k1) > 0
        assert np.array_equal(hd[qofs[i]: qofs[i + 1]], d1) and np.array_equal(hk[qofs[i]: qofs[i + 1]], k1), i
    ok, od, _ = oracle.sift(frames[3], oracle.sift_config(nfeatures=300))
    assert np.array_equal(hd[qofs[3]: qofs[4]], od)
    m.close()


def test_sift_list_capacities_grow_on_demand(capi, oracle, cfg0_data, monkeypatch):
    """The per-frame extrema / keypoint lists have fixed capacities (65536 / 32768: far above what a 4095 x 4095 image yields); a
    frame beyond one re-runs the pass over the same pyramids with that list doubled instead of failing.  Started
    at 64 / 32 entries, every image of the batch overflows several times on the way — same keypoints, same descriptors."""
    pages, frames, _, _ = cfg0_data
    monkeypatch.setenv("SLIDEO_SIFT_LIST_CAP", "64")
    mm = capi.Matcher(capi.default_config())
    assert _cmp(capi, oracle, mm, frames[0]) < 20                  # a "no slide" frame: 13 extrema, nothing grows
    for img in (frames[1], pages[2]):                               # 809 / 2370 extrema, 155 / 221 refined keypoints: both lists grow
        assert _cmp(capi, oracle, mm, img) > 150
    # as a matcher: pages and frames through the grown lists, verdicts equal to the default capacities'
    from conftest import small_cfg
    def run():
        g = capi.Matcher(small_cfg(capi)); g.use_sift(capi.sift_config(nfeatures=300), 0.0)
        g.add_pages(list(pages)); g.finalize()
        v = g.match_frames(frames)
        g.close()
        return v
    va = run()
    monkeypatch.delenv("SLIDEO_SIFT_LIST_CAP")
    assert run().tobytes() == va.tobytes()
    mm.close()
