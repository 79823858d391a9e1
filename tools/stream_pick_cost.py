"""What slideo_matcher_create's stream pick (SLIDEO_STREAM_PICK, include/slideo_amd.h "Environment") costs: create + destroy of a matcher
with and without it, 10 times each (on the GPU box: python tools/stream_pick_cost.py).  r06: 25 - 42 ms against 19 ms."""
import time, sys, os
sys.path.insert(0, '/root/repo')
from slideo_amd import _capi
for rep in range(2):
    for pick in ("1", "0"):
        os.environ["SLIDEO_STREAM_PICK"] = pick
        t0 = time.perf_counter()
        for i in range(10):
            m = _capi.Matcher(_capi.default_config(nfeatures=500)); m.close()
        print("pick", pick, "create+destroy ms", (time.perf_counter() - t0) / 10 * 1e3)
