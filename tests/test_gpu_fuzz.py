"""Randomised parity: HIP path vs the CPU restatement over random image sizes, contents and configurations.

The fixed-shape tests pin the reference's literals and the benchmark shapes; the kernels' edge code (border strips of the
blur, the last groups of a resize row, partial FAST tiles, levels that vanish, ties, tiny decks) depends on sizes modulo 4,
16, 126, 248 ... — a seeded sweep covers combinations nobody wrote down.  Same bar as test_gpu_parity.py: bit-exact keypoints,
angles, descriptors, votes, inliers; |d similarity| <= 1e-4.
"""
import numpy as np
import pytest

from test_gpu_parity import _build_both, _cmp_orb, _compare_traces

pytestmark = pytest.mark.gpu


def _image(rng, synth, w, h):
    kind = rng.integers(0, 4)
    if kind == 0:                                    # blocky noise (many corners, many ties)
        cell = int(rng.integers(3, 9))
        base = rng.integers(0, 256, ((h + cell - 1) // cell, (w + cell - 1) // cell), dtype=np.uint8)
        img = np.kron(base, np.ones((cell, cell), np.uint8))[:h, :w]
        img = np.repeat(img[:, :, None], 3, 2)
    elif kind == 1:                                  # a crop of a synthetic slide
        pg = synth.pages(1, max(w, 320), max(h, 240), seed=int(rng.integers(1, 1 << 30)))[0]
        img = pg[:h, :w]
    elif kind == 2:                                  # smooth gradient + sparse dots (few corners; flat tiles)
        yy, xx = np.mgrid[0:h, 0:w]
        img = ((xx * 255 // max(w - 1, 1)) ^ (yy * 3)).astype(np.uint8)
        img = np.repeat(img[:, :, None], 3, 2).copy()
        for _ in range(60):
            y, x = int(rng.integers(0, h - 6)), int(rng.integers(0, w - 6))
            img[y:y + 5, x:x + 5] = rng.integers(0, 256, 3)
    else:                                            # independent colour noise, quantised
        img = (rng.integers(0, 256, (h, w, 3), dtype=np.uint8) // 32 * 32).astype(np.uint8)
    return np.ascontiguousarray(img)


def _config(rng):
    over = dict(nfeatures=int(rng.choice([60, 300, 1000, 2500])), fast_threshold=int(rng.choice([6, 12, 20, 35])),
                nlevels=int(rng.choice([1, 3, 5, 8])), scale_factor=float(rng.choice([1.1, 1.2, 1.35, 1.7])))
    over["ocv_blur"] = int(rng.integers(0, 4)); over["ocv_gray"] = int(rng.integers(0, 2))
    over["ocv_resize"] = int(rng.integers(0, 2)); over["ocv_atan"] = int(rng.integers(0, 2))
    return over


@pytest.mark.parametrize("seed", range(60))
def test_orb_random_sizes_contents_configs(capi, oracle, synth, seed):
    rng = np.random.default_rng(1000 + seed)
    over = _config(rng)
    m = capi.Matcher(capi.default_config(**over))
    ocfg = oracle.default_config(**over)
    total = 0
    for _ in range(3):
        w, h = int(rng.integers(131, 1000)), int(rng.integers(131, 700))
        total += _cmp_orb(capi, oracle, m, ocfg, _image(rng, synth, w, h))
    m.close()
    assert total >= 0


@pytest.mark.parametrize("seed", range(30))
def test_end_to_end_random_decks(capi, oracle, synth, seed):
    rng = np.random.default_rng(2000 + seed)
    pw, ph = int(rng.integers(500, 1100)), int(rng.integers(300, 700))
    fw, fh = int(rng.integers(400, 1000)), int(rng.integers(260, 640))
    fh = max(fh, 120000 // fw + 1)                                   # (to_small_image only shrinks: area >= small_area, a stated limit)
    npages, nframes = int(rng.integers(2, 7)), int(rng.integers(3, 9))
    pages = synth.pages(npages, pw, ph, seed=int(rng.integers(1, 1 << 30)))
    frames, truth, _ = synth.frames(pages, nframes, fw, fh, seed=int(rng.integers(1, 1 << 30)))
    over = dict(nfeatures=int(rng.choice([300, 800, 1500])), min_rating=float(rng.choice([8.0, 20.0, 50.0])),
                vote_tolerance=float(rng.choice([1.0, 1.05, 1.2])), ocv_blur=int(rng.integers(0, 4)))
    m, db = _build_both(capi, oracle, capi.default_config(**over), oracle.default_config(**over), pages)
    assert m.descriptor_count == db.descriptor_count
    if m.descriptor_count > 0:
        v = m.match_frames(frames)
        _compare_traces(m, db, frames, v)
    m.close()


@pytest.mark.parametrize("seed", range(16))
def test_end_to_end_random_decks_extension_modes(capi, oracle, synth, seed):
    """The same sweep over the round-3 modes: 8-DOF homography verification (both sample solvers, perspective frames), the
    LSH-compatible index, duplicated pages (train-set de-duplication) — traces against the oracle."""
    rng = np.random.default_rng(3000 + seed)
    pw, ph = int(rng.integers(600, 1100)), int(rng.integers(360, 700))
    fw, fh = int(rng.integers(480, 1000)), int(rng.integers(300, 640))
    fh = max(fh, 120000 // fw + 1)
    npages, nframes = int(rng.integers(2, 6)), int(rng.integers(3, 7))
    pages = synth.pages(npages, pw, ph, seed=int(rng.integers(1, 1 << 30)))
    if rng.integers(0, 2):
        pages = np.concatenate([pages, pages[: int(rng.integers(1, npages + 1))]])                  # repeated pages: equal train rows
    mode = seed % 4
    over = dict(nfeatures=int(rng.choice([300, 800, 1500])), min_rating=float(rng.choice([8.0, 20.0])))
    fseed = int(rng.integers(1, 1 << 30))
    if mode in (0, 1):
        over.update(verify_model=1, ocv_hdlt=(mode if seed < 8 else 2), ransac_max_iters=int(rng.choice([200, 2000])), refine_iters=int(rng.choice([0, 10])))
        frames, truth, _ = synth.frames_persp(pages, nframes, fw, fh, persp=float(rng.choice([0.0, 0.1, 0.25])), seed=fseed)
    elif mode == 2:
        over.update(matcher=1, lsh_multi_probe=int(rng.integers(0, 3)), lsh_key_bits=int(rng.choice([8, 12, 14])), lsh_tables=int(rng.integers(1, 8)))
        frames, truth, _ = synth.frames(pages, nframes, fw, fh, seed=fseed)
    else:
        over.update(vote_tolerance=float(rng.choice([1.0, 1.05, 1.2])), knn_k=int(rng.choice([5, 30, 32])))
        frames, truth, _ = synth.frames(pages, nframes, fw, fh, seed=fseed)
    m, db = _build_both(capi, oracle, capi.default_config(**over), oracle.default_config(**over), pages)
    assert m.descriptor_count == db.descriptor_count
    if m.descriptor_count > 0:
        v = m.match_frames(frames)
        _compare_traces(m, db, frames, v, skip_ill_conditioned=mode in (0, 1))
    m.close()


@pytest.mark.parametrize("seed", range(12))
def test_sift_random_sizes_contents_configs(capi, oracle, synth, seed):
    """SIFT (csrc/sift.hip.h) over random sizes / contents / parameters: keypoints and descriptors bit-exact."""
    rng = np.random.default_rng(4000 + seed)
    sc = dict(nfeatures=int(rng.choice([0, 50, 400])), contrast_threshold=float(rng.choice([0.02, 0.04, 0.09])),
              edge_threshold=float(rng.choice([6.0, 10.0, 20.0])), sigma=float(rng.choice([1.2, 1.6, 2.0])))
    over = dict(ocv_blur=int(rng.integers(0, 2)), ocv_gray=int(rng.integers(0, 2)), ocv_atan=int(rng.integers(0, 2)))
    m = capi.Matcher(capi.default_config(**over))
    ocfg = oracle.default_config(**over)
    for _ in range(2):
        w, h = int(rng.integers(40, 700)), int(rng.integers(40, 500))
        img = _image(rng, synth, max(w, 131), max(h, 131))[:h, :w].copy()
        gk, gd = m.sift(img, capi.sift_config(**sc))
        ok, od, _ = oracle.sift(img, oracle.sift_config(**sc), ocfg)
        assert len(gk) == len(ok), (w, h, sc)
        assert np.array_equal(gk, ok.view(gk.dtype)) and np.array_equal(gd, od), (w, h, sc)
    m.close()


@pytest.mark.parametrize("seed", range(8))
def test_sift_matcher_random_decks(capi, oracle, synth, seed):
    """The SIFT matcher mode (slideo_matcher_use_sift) over random deck / frame sizes, SIFT parameters, ratios and both
    verifiers: page features bit-exact, traces against the oracle in the same mode."""
    from test_gpu_sift_matcher import _build
    rng = np.random.default_rng(5000 + seed)
    pw, ph = int(rng.integers(600, 1000)), int(rng.integers(360, 640))
    fw, fh = int(rng.integers(480, 900)), int(rng.integers(300, 560))
    fh = max(fh, 120000 // fw + 1)
    npages, nframes = int(rng.integers(2, 6)), int(rng.integers(2, 5))
    pages = synth.pages(npages, pw, ph, seed=int(rng.integers(1, 1 << 30)))
    if rng.integers(0, 2):
        pages = np.concatenate([pages, pages[:1]])                                                  # a twin page: equal SIFT rows, d1 = d2
    sk = dict(nfeatures=int(rng.choice([0, 150, 400])), contrast_threshold=float(rng.choice([0.04, 0.07])), sigma=float(rng.choice([1.4, 1.6])))
    over = dict(min_rating=float(rng.choice([6.0, 12.0])))
    homog = bool(seed % 2)
    if homog:
        over.update(verify_model=1, ocv_hdlt=int(rng.integers(0, 3)), ransac_max_iters=300)
        frames, truth, _ = synth.frames_persp(pages, nframes, fw, fh, persp=0.1, seed=int(rng.integers(1, 1 << 30)))
    else:
        frames, truth, _ = synth.frames(pages, nframes, fw, fh, seed=int(rng.integers(1, 1 << 30)))
    ratio = float(rng.choice([0.0, 0.6, 0.75, 0.9]))                                                 # 0: the tolerance vote on L2 distances
    if ratio == 0.0:
        over.update(knn_k=int(rng.choice([5, 30, 32])), vote_tolerance=float(rng.choice([1.0, 1.05, 1.2])))
    m, db = _build(capi, oracle, pages, sk, ratio, **over)
    assert m.descriptor_count == db.descriptor_count
    if m.descriptor_count > 1:
        gk, gd = m.page_features(0)
        ok, od = db.page_features(0)
        assert np.array_equal(gd, od) and np.array_equal(gk, ok.view(gk.dtype))
        v = m.match_frames(frames)
        _compare_traces(m, db, frames, v, skip_ill_conditioned=homog)
    m.close()
