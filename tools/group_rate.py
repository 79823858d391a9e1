#!/usr/bin/env python3
"""Host-frames rate of the N-device group (slideo_group_match_frames_bgr8) beside the single matcher's (slideo_match_frames_bgr8):
the headline deck, 1080p frames from pinned host memory, verdicts to host — the PCIe-inclusive path the Rust crate binds
(crates/matching-hip: HipImageVideoMatcher::default() = every gfx950 device of the node).

  python tools/group_rate.py --devices all        one group over every gfx950 device of the node (the library enumerates them:
                                                  slideo_group_create with n_devices 0), 256 frames per device and call — on an
                                                  N-GPU box this ONE command reports the group's PCIe-inclusive rate over N links
  python tools/group_rate.py --devices 0,1,2,3    the same over the named ordinals
  python tools/group_rate.py                      single-GPU box: groups of 1 / 2 / 4 members SHARING device 0 (what is measured is
                                                  the group's overhead: threads, shards, copy streams on one link)
The last line of output is one JSON record (profiles/r05_group_rate.txt keeps the whole output)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (pinned host memory)
from slideo_amd import _capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--devices", default="", help="'all', or comma-separated HIP ordinals (an ordinal may repeat); default: the shared-device sweep")
ap.add_argument("--pages", type=int, default=500)
ap.add_argument("--per-device", type=int, default=256, help="frames per member device and call")
ap.add_argument("--calls", type=int, default=4)
a = ap.parse_args()

if a.devices == "all":
    sweeps = [None]
elif a.devices:
    sweeps = [[int(x) for x in a.devices.split(",")]]
else:
    sweeps = [[0], [0, 0], [0, 0, 0, 0]]
n_max = max(len(_capi.device_list()) if d is None else len(d) for d in sweeps)
per_call = lambda members: a.per_device * (members if a.devices else 1)        # (the shared-device sweep keeps 256 frames per call)
pages = synth.pages(a.pages, 2001, 1125, threads=64)
frames, truth, _ = synth.frames(pages, per_call(n_max), 1920, 1080, threads=64)
pinned = torch.from_numpy(frames).pin_memory().numpy()
cfg = _capi.default_config(nfeatures=1000)


def rate(obj, label, n):
    obj.match_frames(pinned[:n])
    t0 = time.perf_counter()
    for _ in range(a.calls):
        v = obj.match_frames(pinned[:n])
    dt = (time.perf_counter() - t0) / a.calls
    rec = {"what": label, "frames_per_call": n, "ms_per_call": round(dt * 1e3, 1), "frames_per_s": round(n / dt, 1),
           "h2d_inclusive_GBps": round(pinned[:n].nbytes / dt / 1e9, 1), "accuracy": round(float((v["page_idx"] == truth[:n]).mean()), 4)}
    print("%-40s %7.1f ms per %d frames = %7.0f frames/s, %5.1f GB/s H2D-inclusive, accuracy %.3f"
          % (label, rec["ms_per_call"], n, rec["frames_per_s"], rec["h2d_inclusive_GBps"], rec["accuracy"]), flush=True)
    return v, rec


m = _capi.Matcher(cfg, device=_capi.device_list()[0])
for i in range(0, a.pages, 50):
    m.add_pages(list(pages[i:i + 50]))
m.finalize()
ref, single = rate(m, "single matcher (first gfx950 device)", per_call(1))
m.close()
out = {"single_matcher": single, "groups": []}
for members in sweeps:
    t0 = time.perf_counter()
    g = _capi.Group(cfg, devices=members)
    nm = len(g.devices)
    for i in range(0, a.pages, 50 * nm):
        g.add_pages(list(pages[i:i + 50 * nm]))
    g.finalize()
    t_db = time.perf_counter() - t0
    n = per_call(nm)
    v, rec = rate(g, "group over devices %s" % g.devices, n)
    rec.update(devices=g.devices, page_db_over_the_group_s=round(t_db, 2),
               first_shard_equals_single_matcher=bool(v[:min(len(v), len(ref))].tobytes() == ref[:min(len(v), len(ref))].tobytes()),
               speedup_over_single_matcher=round(rec["frames_per_s"] / single["frames_per_s"], 2))
    print("    page DB over the group: %.2f s; first shard's verdicts equal the single matcher's: %s"
          % (t_db, rec["first_shard_equals_single_matcher"]), flush=True)
    out["groups"].append(rec)
    g.close()
print(json.dumps(out))
