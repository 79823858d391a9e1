#!/usr/bin/env python3
"""Concurrency picture of an overlapped bench run from a rocprofv3 rocpd database (--kernel-trace): over the window of the middle
search launches, the share of time with a search kernel running, with 2 running, with none; the same for the ORB and verify groups;
and a coarse timeline (one character per <res> us: K = search, o = ORB only, v = verify only, + = ORB and verify, . = idle).

usage: python tools/timeline_overlap.py <results.db> [res_us]"""
import re, sqlite3, sys

ORB = ("gray_kernel", "resize_kernel", "fast_kernel", "threshold_kernel", "scan_kernel", "compact_kernel", "sort_kernel", "blur_", "describe")
VER = ("vote_kernel", "ransac", "rate_kernel", "reproject", "verdict", "expand_dups")


def group(name):
    if "knn_tile" in name and "expand" not in name: return "K"
    if any(k in name for k in ORB): return "o"
    if any(k in name for k in VER): return "v"
    return None


def main(path, res=100.0):
    c = sqlite3.connect(path)
    rows = [(group(re.sub(r"\(.*", "", n)), s, s + d) for n, s, d in c.execute("select name, start, duration from kernels order by start")]
    ks = [(s, e) for g, s, e in rows if g == "K"]
    if len(ks) < 6: print("too few search launches"); return
    lo, hi = ks[len(ks) // 4][0], ks[(3 * len(ks)) // 4][1]      # the middle half of the search launches: steady state (no warm-up, no drain, not bench.py's one-batch-in-flight tail)
    ev = []
    for g, s, e in rows:
        if g and e > lo and s < hi: ev.append((max(s, lo), 1, g)); ev.append((min(e, hi), -1, g))
    ev.sort()
    cnt = {"K": 0, "o": 0, "v": 0}
    acc = {}
    t = lo
    for tt, d, g in ev:
        key = (min(cnt["K"], 2), cnt["o"] > 0, cnt["v"] > 0)
        acc[key] = acc.get(key, 0) + (tt - t); t = tt
        cnt[g] += d
    tot = float(hi - lo)
    print("window %.2f ms, %d search launches inside" % (tot / 1e6, sum(1 for s, e in ks if s >= lo and e <= hi)))
    for k in sorted(acc): print("  search running x%d  orb %-5s verify %-5s  %5.1f %%" % (k[0], k[1], k[2], 100 * acc[k] / tot))
    # timeline
    n = int(tot / (res * 1e3)) + 1
    line = []
    for i in range(n):
        a, b = lo + i * res * 1e3, lo + (i + 1) * res * 1e3
        g = {x for x, s, e in rows if x and e > a and s < b}
        line.append("K" if "K" in g and len(g) == 1 else "#" if "K" in g else "+" if g == {"o", "v"} else "o" if g == {"o"} else "v" if g == {"v"} else ".")
    s = "".join(line)
    print("timeline (%g us per char; K search only, # search + others, o ORB, v verify, + both, . idle)" % res)
    for i in range(0, len(s), 120): print("  " + s[i:i + 120])


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 100.0)
