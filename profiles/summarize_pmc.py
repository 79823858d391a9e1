#!/usr/bin/env python3
"""Per-kernel PMC sums from a rocprofv3 --pmc rocpd database.  usage: summarize_pmc.py <db> [kernel substring]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(list)
for name, cn, val, d, did in c.execute("select kernel_name, counter_name, value, duration, dispatch_id from counters_collection"):
    if pat in name:
        k = name.split("(")[0][:48]; acc[k][cn] += val; dur[(k, did)] = d
for k, cs in acc.items():
    ds = [d for (kk, _), d in dur.items() if kk == k]
    print("%s: dispatches %d, total %.1f us" % (k, len(ds), sum(ds) / 1e3))
    for cn, v in sorted(cs.items()): print("    %-28s %16.0f" % (cn, v))
