#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max, share) of a rocprofv3 rocpd database.

usage: python profiles/summarize_rocpd.py <results.db> [> profiles/<name>.txt]
(rocprofv3 --kernel-trace --stats writes a rocpd SQLite database on ROCm 7.2; this prints what
the CSV kernel_stats file of older versions held.)
"""
import re
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, duration, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, sgpr_count, lds_size "
                     "from kernels order by start").fetchall()
    agg = {}
    for name, dur, gx, gy, gz, wx, vg, sg, lds in rows:
        short = re.sub(r"\(.*", "", name)
        a = agg.setdefault(short, dict(n=0, tot=0, mn=1 << 62, mx=0, vg=vg, sg=sg, lds=lds, grid=(gx, gy, gz), wg=wx))
        a["n"] += 1; a["tot"] += dur; a["mn"] = min(a["mn"], dur); a["mx"] = max(a["mx"], dur)
    total = sum(a["tot"] for a in agg.values()) or 1
    print("%-44s %6s %12s %12s %12s %12s %6s  %5s %5s %7s  %s" %
          ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%", "vgpr", "sgpr", "lds", "last grid(threads) / wg"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        print("%-44s %6d %12.1f %12.2f %12.2f %12.2f %6.2f  %5s %5s %7s  %s / %s" %
              (k[:44], a["n"], a["tot"] / 1e3, a["tot"] / a["n"] / 1e3, a["mn"] / 1e3, a["mx"] / 1e3,
               100.0 * a["tot"] / total, a["vg"], a["sg"], a["lds"], "x".join(map(str, a["grid"])), a["wg"]))
    print("total kernel time: %.1f us over %d dispatches" % (total / 1e3, len(rows)))
    # the dominant kernel launch by launch, in start order: bench.py issues <warmup> launches, <steps> timed ones and
    # (two-stream mode only) 3 more with one batch in flight after the timed region
    dom = max((kv for kv in agg.items() if "rocclr" not in kv[0]), key=lambda kv: kv[1]["tot"], default=(None, None))[0]
    durs = [dur / 1e3 for name, dur, *_ in rows if re.sub(r"\(.*", "", name) == dom]
    print("%s launches in start order (us): %s" % (dom, " ".join("%.0f" % d for d in durs)))


if __name__ == "__main__":
    main(sys.argv[1])
