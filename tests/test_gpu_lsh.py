"""slideo_config.matcher 1 (the LSH-compatible search, csrc/knn_lsh.hip.h) against the CPU restatement, through the C ABI:
neighbour lists bit-exact, end-to-end traces equal."""
import numpy as np
import pytest

from conftest import small_cfg
from test_gpu_parity import _build_both, _compare_traces

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mp,kb,ntab", [(1, 12, 6), (0, 12, 6), (2, 10, 3), (1, 16, 8), (1, 4, 2)])
def test_knn_lsh_bit_exact(capi, oracle, mp, kb, ntab):
    rng = np.random.default_rng(7)
    base = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    t = base[rng.integers(0, 60, 20000)].copy()
    for i in range(len(t)):
        for p in rng.integers(0, 256, rng.integers(0, 5)):
            t[i, p >> 3] ^= 1 << (p & 7)
    q = np.concatenate([base[rng.integers(0, 60, 300)], rng.integers(0, 256, (50, 32), dtype=np.uint8)])
    t = np.concatenate([t, rng.integers(0, 256, (40, 32), dtype=np.uint8)])          # + a few isolated rows
    kw = dict(matcher=1, lsh_multi_probe=mp, lsh_key_bits=kb, lsh_tables=ntab)
    m = capi.Matcher(capi.default_config(**kw))
    gi, gd = m.knn_lsh(q, t, 30)
    oi, od, _ = oracle.knn_lsh(q, t, 30, oracle.default_config(**kw))
    assert np.array_equal(gd, od) and np.array_equal(gi, oi)
    assert (gi >= 0).any()
    m.close()


def test_knn_lsh_many_ties_at_the_kth_distance(capi, oracle):
    """2500 copies of one descriptor: every candidate at distance 0 — far more ties than the LDS tie list holds (the repeated
    minimum passes), and the 30 lowest rows must come back."""
    rng = np.random.default_rng(3)
    one = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    t = np.concatenate([rng.integers(0, 256, (500, 32), dtype=np.uint8), np.repeat(one, 2500, 0), rng.integers(0, 256, (300, 32), dtype=np.uint8)])
    kw = dict(matcher=1)
    m = capi.Matcher(capi.default_config(**kw))
    gi, gd = m.knn_lsh(one, t, 30)
    oi, od, _ = oracle.knn_lsh(one, t, 30, oracle.default_config(**kw))
    assert np.array_equal(gi, oi) and np.array_equal(gd, od) and list(gi[0]) == list(range(500, 530)) and not gd.any()
    m.close()


@pytest.mark.parametrize("share", [None, "4", "6"])
def test_lsh_mode_traces(capi, oracle, synth, cfg0_data, monkeypatch, share):
    """(share: SLIDEO_KNN_SHARE forcing the 12-wave block shapes of the EXACT search — the LSH-filtered stream keeps its own 8-wave
    kernel and plan, and its buffers must be reserved for that plan: ADVICE r05, stage_knn.hip knn_shape)"""
    if share is not None:
        monkeypatch.setenv("SLIDEO_KNN_SHARE", share)
    pages, frames, truth, _ = cfg0_data
    m, db = _build_both(capi, oracle, small_cfg(capi, matcher=1), small_cfg(oracle, matcher=1), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v)
    m.close()
    pages = synth.pages(48, 800, 450)
    frames, truth, _ = synth.frames(pages, 24, 640, 360)
    m, db = _build_both(capi, oracle, small_cfg(capi, matcher=1), small_cfg(oracle, matcher=1), pages)
    v = m.match_frames(frames)
    _compare_traces(m, db, frames, v)
    assert (v["page_idx"] == truth).mean() >= 0.8
    big = np.concatenate([frames] * 6)                      # pipelined units (capacity-sized grids, device-side query counts)
    assert np.array_equal(m.match_frames(big), np.concatenate([v] * 6))
    m.close()
    with pytest.raises(capi.SlideoError) as e:
        capi.Matcher(capi.default_config(matcher=1, lsh_key_bits=17))
    assert e.value.code == 5
