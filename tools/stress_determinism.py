"""Repeats the headline batch many times through the streaming entry points (two units in flight) and checks that every
repetition returns bit-identical verdict records and candidate traces — a race in the kNN ring protocol, the pending
buffers or the slot pipeline would show up as a difference.  usage (GPU box): python tools/stress_determinism.py [reps] [mode]
mode "sift": the SIFT matcher mode; mode "homography": verify_model 1, ocv.hdlt 1 on perspective frames — ransac_h_tail_kernel's hand-over list, refine_h's
eigenproblem list and the lane LM are filled through atomics in whatever order the blocks arrive; the results must not care.
The candidate traces of EVERY repetition's last unit are compared in that mode (the lists differ from run to run)."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch
from slideo_amd import _capi, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mode = sys.argv[2] if len(sys.argv) > 2 else "default"
P, B = 500, 256
pages = synth.pages(P, 2001, 1125, threads=64)
if mode == "sift":
    # the SIFT matcher mode (BASELINE configs[2]): SIFT extraction (LDS-DMA stream blurs), L2 2-NN on the int8 matrix cores, Lowe's
    # ratio test, then the shared verify stages
    frames, truth, _ = synth.frames(pages, B, 1920, 1080, threads=64)
    m = _capi.Matcher(_capi.default_config(nfeatures=1000))
    m.use_sift(_capi.sift_config(nfeatures=1000), 0.8)
elif mode == "homography":
    frames, truth, _ = synth.frames_persp(pages, B, 1920, 1080, persp=0.1, threads=64)
    m = _capi.Matcher(_capi.default_config(nfeatures=1000, verify_model=1, ocv_hdlt=1))
else:
    frames, truth, _ = synth.frames(pages, B, 1920, 1080, threads=64)
    m = _capi.Matcher(_capi.default_config(nfeatures=1000))
for i in range(0, P, 50):
    m.add_pages(list(pages[i:i + 50]))
m.finalize()
d = torch.from_numpy(frames).cuda()
ref = m.match_frames_dev(d.data_ptr(), B, 1920, 1080)
ref_c = [m.last_candidates(i).tobytes() for i in range(B)]
bad = 0
t0 = time.time()
pending = []
for r in range(reps):
    if len(pending) == m.max_in_flight():
        v = m.collect(pending.pop(0))
        bad += int(not np.array_equal(v, ref))
    pending.append(m.submit_dev(d.data_ptr(), B, 1920, 1080))
while pending:
    v = m.collect(pending.pop(0))
    bad += int(not np.array_equal(v, ref))
dt = time.time() - t0
# candidate traces of one more synchronous run (homography mode: of 50 more)
bad_c = 0
for _ in range(50 if mode == "homography" else 1):
    v = m.match_frames_dev(d.data_ptr(), B, 1920, 1080)
    bad_c += sum(int(m.last_candidates(i).tobytes() != ref_c[i]) for i in range(B))
print("mode %s, reps %d: %d differing verdict batches, %d differing candidate traces, accuracy %.4f, %.0f frames/s" %
      (mode, reps, bad, bad_c, float((ref["page_idx"] == truth).mean()), reps * B / dt))
sys.exit(1 if bad or bad_c else 0)
