#!/usr/bin/env python3
"""What can "parity unpinned" change at the VERDICT level?

The reference's arithmetic lives in OpenCV 4.5.2, whose source is not available here; every primitive whose rounding could only be
recalled is a named switch (slideo_ocv_variants, include/slideo_amd.h), implemented on the GPU and in the CPU restatement alike.
Until OpenCV's own outputs pin the switches (tools/pin_opencv.py), the strongest parity statement available is how much the
RESULT of the reference's calls (mo/feature_extractor.rs:32-40 detectAndCompute, mo/image_utils.rs:17 resize(INTER_AREA),
mo/image_utils.rs:52 estimateAffinePartial2D on the matches, mo/lib.rs:339-347 warpAffine + similarity) moves when a switch is
set to its alternative: this script measures that at the HEADLINE size — 256 synthetic 1080p frames against the 500-page deck,
ORB-1000 — and on the reference's three real fixture frames (tests/golden: data/matchings/test1), one switch at a time, pages
and frames both analysed under the switch (as a different OpenCV build would).

Per switch, against the default configuration:
  verdicts_changed          frames whose page (or "none") differs, of 256; and the accuracy against the synthetic truth
  candidates                candidate (frame, page) pairs present in both runs; those whose RANSAC inlier count differs, the largest
                            difference, those whose survival of the rating filter (mo/lib.rs:333) differs
  max_similarity_difference over the candidates that survive in both runs (re-projection similarity, mo/lib.rs:351)
  descriptors               on a 16-frame sample (the ORB tap): keypoints present in both runs (same level, x, y), the fraction of
                            their 256-bit descriptors that differ at all, and the bit flip rate among all their bits
The GPU results are what is compared; tests/test_gpu_variant_sensitivity.py holds GPU == CPU restatement on a 16-frame sample of the
same workload under each switch.

    python tools/variant_sensitivity.py > profiles/r05_variant_sensitivity.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SWITCHES = [("default", {}), ("gray 1", dict(ocv_gray=1)), ("blur 1", dict(ocv_blur=1)), ("blur 2", dict(ocv_blur=2)),
            ("blur 3", dict(ocv_blur=3)), ("resize 1", dict(ocv_resize=1)), ("atan 1", dict(ocv_atan=1)), ("area 1", dict(ocv_area=1))]
WHAT = {"gray 1": "cvtColor BGR2GRAY with Q14 coefficients (OpenCV 2.4 / 3.x) instead of Q15",
        "blur 1": "ORB's GaussianBlur as f32 sepFilter2D WITHOUT fma contraction (no AVX2 dispatch)",
        "blur 2": "the Q8 integer sepFilter2D of OpenCV < 4.2 (taps sum to 257)",
        "blur 3": "GaussianBlur's bit-exact fixed-point path (what a non-submatrix source takes)",
        "resize 1": "INTER_LINEAR_EXACT coefficient rounding: floor(x + 0.5) instead of ties-to-even",
        "atan 1": "fastAtan2's polynomial contracted to fma",
        "area 1": "INTER_AREA with exact box weights (no 1e-3 edge cut-off)"}


def run(capi, pages, frames, over, nfeatures, sample):
    m = capi.Matcher(capi.default_config(nfeatures=nfeatures, **over))
    for i in range(0, len(pages), 50):
        m.add_pages(list(pages[i:i + 50]))
    m.finalize()
    v = m.match_frames(frames)
    cands = [m.last_candidates(i) for i in range(len(frames))]
    feats = {i: m.orb(frames[i]) for i in sample}
    M = m.descriptor_count
    m.close()
    return {"v": v, "c": cands, "f": feats, "M": M}


def compare(a, b, truth):
    """b (a switch) against a (the default)"""
    va, vb = a["v"], b["v"]
    n_both = inl_diff = surv_diff = 0
    worst = 0
    dsim = 0.0
    set_diff = 0
    for x, y in zip(a["c"], b["c"]):
        px = {int(c["page_idx"]): c for c in x}
        py = {int(c["page_idx"]): c for c in y}
        set_diff += len(set(px) ^ set(py))
        for p in set(px) & set(py):
            n_both += 1
            d = abs(int(px[p]["inliers"]) - int(py[p]["inliers"]))
            inl_diff += d != 0
            worst = max(worst, d)
            surv_diff += int(px[p]["survived"]) != int(py[p]["survived"])
            if px[p]["survived"] and py[p]["survived"]:
                dsim = max(dsim, abs(float(px[p]["similarity"]) - float(py[p]["similarity"])))
    kp_both = kp_only = desc_diff = bits = 0
    for i in a["f"]:
        (ka, da), (kb, db) = a["f"][i], b["f"][i]
        key = lambda k: {(int(o), float(x), float(y)): j for j, (o, x, y) in enumerate(zip(k["octave"], k["x"], k["y"]))}
        ia, ib = key(ka), key(kb)
        common = sorted(set(ia) & set(ib))
        kp_both += len(common)
        kp_only += len(set(ia) ^ set(ib))
        if common:
            xa = da[[ia[c] for c in common]]; xb = db[[ib[c] for c in common]]
            x = np.unpackbits(xa ^ xb, axis=1)
            desc_diff += int((x.sum(1) > 0).sum())
            bits += int(x.sum())
    changed = va["page_idx"] != vb["page_idx"]
    return {"verdicts_changed": int(changed.sum()), "frames": int(len(va)),
            "verdicts_changed_between_pages": int((changed & (va["page_idx"] >= 0) & (vb["page_idx"] >= 0)).sum()),
            "accuracy_vs_truth": round(float((vb["page_idx"] == truth).mean()), 4) if truth is not None else None,
            "train_descriptors": int(b["M"]),
            "candidates_in_both_runs": n_both, "candidate_pages_in_one_run_only": set_diff,
            "candidates_with_another_inlier_count": int(inl_diff), "largest_inlier_count_difference": int(worst),
            "candidates_with_another_survival": int(surv_diff), "max_similarity_difference": round(dsim, 6),
            "descriptors": {"sample_frames": len(a["f"]), "keypoints_in_both_runs": kp_both, "keypoints_in_one_run_only": kp_only,
                            "descriptors_that_differ": desc_diff,
                            "fraction_of_descriptors_that_differ": round(desc_diff / max(kp_both, 1), 5),
                            "bit_flip_rate": round(bits / max(kp_both * 256, 1), 6)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--pages", type=int, default=500)
    ap.add_argument("--nfeatures", type=int, default=1000)
    a = ap.parse_args()
    from PIL import Image
    from slideo_amd import _capi, synth
    ncpu = os.cpu_count() or 1
    pages = synth.pages(a.pages, threads=min(64, ncpu))
    frames, truth, _ = synth.frames(pages, a.frames, 1920, 1080, threads=min(64, ncpu))
    sample = sorted(set(int(x) for x in np.linspace(0, a.frames - 1, 16)))
    G = os.path.join(ROOT, "tests", "golden")
    load = lambda n: np.ascontiguousarray(np.array(Image.open(os.path.join(G, n)).convert("RGB"))[:, :, ::-1])
    rpages = [load("1-slide.png"), load("3-slide.png")]
    rframes = np.stack([load("1-frame.png"), load("2-frame.png"), load("3-frame.png")])
    out = {"workload": "%d synthetic 1080p frames vs %d pages, ORB-%d, reference defaults otherwise; one slideo_ocv_variants switch at a time, "
                       "pages and frames analysed under the switch; GPU results" % (a.frames, a.pages, a.nfeatures),
           "real_fixtures": "tests/golden (the reference's data/matchings/test1): 3 frames vs 2 slides, ORB-2000 (reference literals); implied verdicts 0, -1, 1",
           "switches": {}}
    base = real0 = None
    for name, over in SWITCHES:
        r = run(_capi, pages, frames, over, a.nfeatures, sample)
        rr = run(_capi, rpages, rframes, over, 2000, [0, 2])
        if base is None:
            base, real0 = r, rr
            out["default"] = {"accuracy_vs_truth": round(float((r["v"]["page_idx"] == truth).mean()), 4), "train_descriptors": int(r["M"]),
                              "real_fixture_verdicts": [int(x) for x in rr["v"]["page_idx"]],
                              "real_fixture_inliers": [int(x) for x in rr["v"]["inliers"]]}
            continue
        rec = compare(base, r, truth)
        rec["what"] = WHAT[name]
        rf = compare(real0, rr, None)
        rec["real_fixtures"] = {"verdicts": [int(x) for x in rr["v"]["page_idx"]], "verdicts_changed": rf["verdicts_changed"],
                                "inliers": [int(x) for x in rr["v"]["inliers"]],
                                "candidates_with_another_inlier_count": rf["candidates_with_another_inlier_count"],
                                "largest_inlier_count_difference": rf["largest_inlier_count_difference"],
                                "max_similarity_difference": rf["max_similarity_difference"],
                                "bit_flip_rate": rf["descriptors"]["bit_flip_rate"]}
        out["switches"][name] = rec
        print("%-9s verdicts changed %3d / %d  candidates with another inlier count %5d / %d  max |d sim| %.4f  bit flips %.4f %%"
              % (name, rec["verdicts_changed"], rec["frames"], rec["candidates_with_another_inlier_count"], rec["candidates_in_both_runs"],
                 rec["max_similarity_difference"], 100 * rec["descriptors"]["bit_flip_rate"]), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
