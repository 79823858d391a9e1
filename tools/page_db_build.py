"""Page-DB build: every rank analysing the whole deck (what bench.py does) versus the page-sharded build of SURVEY.md 8e
(slideo_amd.distributed.build_page_db_sharded).  One GPU here, so the sharded build is measured in its parts: a rank's
share of the analysis, the export of its records, the import of all records, and the bytes the all-gather would move.

    python tools/page_db_build.py [--pages 500] [--world 8] > profiles/r02_page_db_build.json
"""
import argparse, json, os, pickle, sys, time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from slideo_amd import _capi as capi
from slideo_amd import distributed as D
from slideo_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=500)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    a = ap.parse_args()
    pages = synth.pages(a.pages, a.width, a.height)
    mk = lambda: capi.Matcher(capi.default_config())

    def analyse(lo, hi):
        m = mk()
        for i in range(lo, hi, 50):
            m.add_pages(list(pages[i:min(hi, i + 50)]))
        return m

    analyse(0, 8).close()                                                     # warm
    torch.cuda.synchronize(); t = time.perf_counter()
    full = analyse(0, a.pages); t_analyse_all = time.perf_counter() - t
    t = time.perf_counter(); full.finalize(); t_finalize = time.perf_counter() - t
    lo, hi = D.shard_range(a.pages, 0, a.world)
    t = time.perf_counter(); part = analyse(lo, hi); t_analyse_share = time.perf_counter() - t
    t = time.perf_counter()
    recs = []
    for j in range(hi - lo):
        kp, desc = part.page_features(j)
        recs.append((a.width, a.height, kp, desc, part.page_small(j)))
    t_export = time.perf_counter() - t
    share_bytes = len(pickle.dumps(recs, protocol=4))
    allrecs = []
    for j in range(a.pages):
        kp, desc = full.page_features(j)
        allrecs.append((a.width, a.height, kp, desc, full.page_small(j)))
    t = time.perf_counter()
    imp = mk()
    for w, h, kp, desc, small in allrecs:
        imp.add_page_features(w, h, kp, desc, small)
    t_import = time.perf_counter() - t
    t = time.perf_counter(); imp.finalize(); t_finalize_imp = time.perf_counter() - t
    same = imp.descriptor_count == full.descriptor_count
    frames, truth, _ = synth.frames(pages, 16, 1280, 720)
    same = same and bool(np.array_equal(imp.match_frames(frames), full.match_frames(frames)))
    print(json.dumps({
        "pages": a.pages, "page_size": [a.width, a.height], "world": a.world, "descriptors": int(full.descriptor_count),
        "redundant_build_s": {"analyse_all_pages": round(t_analyse_all, 3), "finalize": round(t_finalize, 3)},
        "sharded_build_s": {"analyse_own_share": round(t_analyse_share, 3), "export_own_records": round(t_export, 3),
                            "import_all_records": round(t_import, 3), "finalize": round(t_finalize_imp, 3)},
        "all_gather_bytes_per_rank": share_bytes, "all_gather_bytes_total": share_bytes * a.world,
        "note": "the all-gather itself (RCCL all_gather_object of ~%d MB per rank) is not timed on one GPU" % (share_bytes >> 20),
        "imported_db_equals_direct_db": same,
    }))


if __name__ == "__main__":
    main()
