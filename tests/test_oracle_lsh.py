"""slideo_config.matcher 1 in the CPU restatement: the candidate rule of FLANN's LshIndex as the reference configures it
(crates/matching-opencv/src/flann.rs:14-26).  Definitions pinned against numpy."""
import numpy as np

from conftest import small_cfg


def _keys(t, bits):
    b = np.unpackbits(t, axis=1, bitorder="little")                      # bit position p = byte * 8 + bit
    return [(b[:, pos].astype(np.int64) << np.arange(len(pos))).sum(1) for pos in bits]


def test_lsh_tables_and_candidate_rule(oracle):
    rng = np.random.default_rng(0)
    cfg = oracle.default_config(matcher=1)
    assert (cfg.lsh_tables, cfg.lsh_key_bits, cfg.lsh_multi_probe) == (6, 12, 1)           # mo/flann.rs:16-18
    base = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    t = base[rng.integers(0, 40, 6000)].copy()
    flips = rng.integers(0, 256, (6000, 3))
    for i in range(6000):
        for p in flips[i][: rng.integers(0, 4)]:
            t[i, p >> 3] ^= 1 << (p & 7)
    q = base[rng.integers(0, 40, 50)].copy()
    idx, dist, bits = oracle.knn_lsh(q, t, 30, cfg)
    assert bits.shape == (6, 12) and all(len(set(r)) == 12 and list(r) == sorted(r) for r in bits.tolist()) and bits.min() >= 0 and bits.max() < 256
    # deterministic tables (cv::RNG's default state), different from table to table
    assert np.array_equal(bits, oracle.knn_lsh(q[:1], t[:10], 1, cfg)[2]) and len({tuple(r) for r in bits.tolist()}) == 6
    # numpy restatement of the rule: candidate iff some table's key differs from the query's in <= 1 bit; k best (distance, row)
    tk, qk = _keys(t, bits), _keys(q, bits)
    d_all = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
    for i in range(len(q)):
        cand = np.zeros(len(t), bool)
        for a, b in zip(tk, qk):
            x = a ^ b[i]
            cand |= (x & (x - 1)) == 0                                     # zero or one differing bit
        rows = np.flatnonzero(cand)
        order = rows[np.lexsort((rows, d_all[i, rows]))][:30]
        want_i = np.full(30, -1); want_i[: len(order)] = order
        assert np.array_equal(idx[i], want_i), i
        assert np.array_equal(dist[i][: len(order)], d_all[i, order])
    # recall against the exact search: high for near-duplicates, not 1 in general
    ei, ed = oracle.knn_hamming(q, t, 30)
    assert (dist[:, 0] == ed[:, 0]).mean() > 0.9
    # multi-probe 0 probes fewer buckets, 2 more: candidate sets nest
    n = []
    for mp in (0, 1, 2):
        ii, dd, _ = oracle.knn_lsh(q, t, 30, oracle.default_config(matcher=1, lsh_multi_probe=mp))
        n.append((ii >= 0).sum())
    assert n[0] <= n[1] <= n[2]


def test_lsh_mode_end_to_end_cfg0(oracle, cfg0_data):
    """The whole frame path on the reference's approximate index: same pages as the exact search on these easy frames."""
    pages, frames, truth, _ = cfg0_data
    res = {}
    for mode in (0, 1):
        db = oracle.PageDB(small_cfg(oracle, matcher=mode))
        db.add_pages(pages, threads=4)
        assert db.finalize() == 0
        res[mode] = db.match_frames(frames, threads=4)
    assert list(res[0]["page_idx"]) == list(truth)
    assert (res[1]["page_idx"] == truth).mean() >= 0.75 and all(g == t or g == -1 for g, t in zip(res[1]["page_idx"], truth))
