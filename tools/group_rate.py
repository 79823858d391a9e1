#!/usr/bin/env python3
"""Host-frames rate of the N-device group (slideo_group_match_frames_bgr8) beside the single matcher's (slideo_match_frames_bgr8):
the headline deck, 256 x 1080p frames per call from pinned host memory, verdicts to host — the PCIe-inclusive path the Rust crate
binds.  On a single-GPU box the members share device 0 (what is measured is the group's overhead: threads, shards, two copy
streams on one link); on a node with several GPUs pass their ordinals.   usage: group_rate.py [ordinals, e.g. 0,1,2,3]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (pinned host memory)
from slideo_amd import _capi, synth  # noqa: E402

devs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else None
pages = synth.pages(500, 2001, 1125, threads=64)
frames, truth, _ = synth.frames(pages, 256, 1920, 1080, threads=64)
pinned = torch.from_numpy(frames).pin_memory().numpy()
cfg = _capi.default_config(nfeatures=1000)


def rate(obj, label, n=256):
    obj.match_frames(pinned[:n])
    t0 = time.perf_counter()
    for _ in range(4):
        v = obj.match_frames(pinned[:n])
    dt = (time.perf_counter() - t0) / 4
    print("%-34s %6.1f ms per %d frames = %6.0f frames/s, %5.1f GB/s H2D-inclusive, accuracy %.3f"
          % (label, dt * 1e3, n, n / dt, pinned[:n].nbytes / dt / 1e9, float((v["page_idx"] == truth[:n]).mean())))
    return v


m = _capi.Matcher(cfg)
for i in range(0, 500, 50):
    m.add_pages(list(pages[i:i + 50]))
m.finalize()
ref = rate(m, "single matcher (device 0)")
m.close()
m = _capi.Matcher(cfg)                      # (a second matcher of the process: the first one's buffers went back to the allocator)
for i in range(0, 500, 50):
    m.add_pages(list(pages[i:i + 50]))
m.finalize()
rate(m, "single matcher, created second")
m.close()
for members in ([[0], [0, 0], [0, 0, 0, 0]] if devs is None else [devs]):
    t0 = time.perf_counter()
    g = _capi.Group(cfg, devices=members)
    for i in range(0, 500, 50 * len(members)):
        g.add_pages(list(pages[i:i + 50 * len(members)]))
    g.finalize()
    t_db = time.perf_counter() - t0
    v = rate(g, "group over devices %s" % members)
    print("    page DB over the group: %.2f s; verdicts equal the single matcher's: %s" % (t_db, bool(v.tobytes() == ref.tobytes())))
    g.close()
