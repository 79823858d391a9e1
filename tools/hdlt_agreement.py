#!/usr/bin/env python3
"""verify_model 1: how far do the three 4-point sample solvers (slideo_ocv_variants.hdlt 0 / 1 / 2) agree END TO END?

hdlt 0 is cv::findHomography's own form (9 x 9 L^T L, Jacobi eigenvectors); 1 (8 x 8 elimination) and 2 (closed form) give the same
model of a sample to f64 round-off at 1 / 5 .. 1 / 60 of the cost.  Round-off can flip an inlier at the 3 px threshold, and RANSAC's
adaptive stop then takes another path — this script counts how often that happens on the headline shape: 256 perspective
1080p frames against the 500-page deck, every candidate's inlier count and every verdict, form against form.  The GPU results
are what is compared (each form is held bit-exact to the CPU restatement of the same form by tests/test_gpu_big_shapes.py).

    python tools/hdlt_agreement.py [--frames 256] [--pages 500] > profiles/r04_hdlt_agreement.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(pages, frames, hdlt, nfeatures):
    from slideo_amd import _capi
    m = _capi.Matcher(_capi.default_config(nfeatures=nfeatures, verify_model=1, ocv_hdlt=hdlt))
    for i in range(0, len(pages), 50):
        m.add_pages(list(pages[i:i + 50]))
    m.finalize()
    v = m.match_frames(frames)
    cands = [m.last_candidates(i) for i in range(len(frames))]
    m.close()
    return v, cands


def compare(a, b):
    (va, ca), (vb, cb) = a, b
    n_c = same_pages = same_inl = same_surv = 0
    worst = 0
    for x, y in zip(ca, cb):
        n_c += len(x)
        if list(x["page_idx"]) == list(y["page_idx"]):
            same_pages += len(x)
            eq = x["inliers"] == y["inliers"]
            same_inl += int(eq.sum())
            same_surv += int((x["survived"] == y["survived"]).sum())
            if len(x):
                worst = max(worst, int(np.abs(x["inliers"] - y["inliers"]).max()))
    return {"frames": len(va), "verdict_page_agreement": float((va["page_idx"] == vb["page_idx"]).mean()),
            "max_similarity_difference": float(np.abs(va["similarity"] - vb["similarity"]).max()),
            "candidates": n_c, "candidate_inlier_count_agreement": same_inl / max(n_c, 1),
            "candidate_survival_agreement": same_surv / max(n_c, 1), "largest_inlier_count_difference": worst}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--pages", type=int, default=500)
    ap.add_argument("--nfeatures", type=int, default=1000)
    a = ap.parse_args()
    from slideo_amd import synth
    ncpu = os.cpu_count() or 1
    pages = synth.pages(a.pages, threads=min(64, ncpu))
    frames, truth, _ = synth.frames_persp(pages, a.frames, 1920, 1080, persp=0.1, threads=min(64, ncpu))
    res = {h: run(pages, frames, h, a.nfeatures) for h in (0, 1, 2)}
    out = {"workload": "%d perspective 1080p frames (projective component 0.1) vs %d pages, ORB-%d, verify_model 1" % (a.frames, a.pages, a.nfeatures),
           "accuracy_vs_truth": {str(h): float((res[h][0]["page_idx"] == truth).mean()) for h in res},
           "hdlt1_vs_hdlt0": compare(res[1], res[0]), "hdlt2_vs_hdlt0": compare(res[2], res[0])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
