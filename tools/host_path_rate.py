import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slideo_amd import _capi, synth
pages = synth.pages(500, 2001, 1125, threads=64)
frames, truth, _ = synth.frames(pages, 256, 1920, 1080, threads=64)
m = _capi.Matcher(_capi.default_config(nfeatures=1000))
for i in range(0, 500, 50): m.add_pages(list(pages[i:i+50]))
m.finalize()
pinned = torch.from_numpy(frames).pin_memory().numpy()
for name, arr in (("pageable", frames), ("pinned", pinned)):
    m.match_frames(arr)
    t0 = time.perf_counter()
    for _ in range(3): v = m.match_frames(arr)
    dt = (time.perf_counter() - t0) / 3
    print("host frames (%s): %.1f ms per 256 frames = %.0f frames/s, %.1f GB/s H2D-inclusive" % (name, dt * 1e3, 256 / dt, frames.nbytes / dt / 1e9))
