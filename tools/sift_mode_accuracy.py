"""Accuracy of the SIFT matcher mode (slideo_matcher_use_sift) on the configs[2] workload against the synthetic truth, over Lowe's
ratio and the reference's absolute rating threshold (mo/lib.rs:338, min_rating).  usage (GPU box): python tools/sift_mode_accuracy.py"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slideo_amd import _capi, synth
P, B = 500, 256
pages = synth.pages(P, 2001, 1125, threads=64)
frames, truth, _ = synth.frames(pages, B, 1920, 1080, threads=64)
d = torch.from_numpy(frames).cuda()
for ratio in (0.75, 0.85, 0.0):                # 0.0 = the path's tolerance vote on the L2 distances (knn_k = 30)
    for mr in (50.0, 25.0, 12.0):
        m = _capi.Matcher(_capi.default_config(min_rating=mr))
        m.use_sift(_capi.sift_config(nfeatures=1000), ratio)
        for i in range(0, P, 50): m.add_pages(list(pages[i:i+50]))
        m.finalize()
        v = m.match_frames_dev(d.data_ptr(), B, 1920, 1080)
        wrong = int(((v["page_idx"] != truth) & (v["page_idx"] >= 0) & (truth >= 0)).sum())
        none_t = int((truth < 0).sum()); fp = int(((truth < 0) & (v["page_idx"] >= 0)).sum())
        print("ratio %.2f min_rating %4.0f: accuracy %.4f, wrong page %d, false positives on no-slide frames %d/%d, median inliers of hits %.0f" % (
            ratio, mr, float((v["page_idx"] == truth).mean()), wrong, fp, none_t, float(np.median(v["inliers"][v["page_idx"] >= 0])) if (v["page_idx"] >= 0).any() else 0))
        m.close()
