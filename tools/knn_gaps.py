"""usage: python tools/knn_gaps.py <rocpd results.db>  -> how the kNN launches tile the timeline of a bench run:
per launch start/end (ms from the first), gap to the previous launch's end (negative = overlap), and over the timed
steps the share of wall time with 0 / 1 / 2+ kNN launches resident."""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
knn = [(s, e) for n, s, e in rows if "knn_tile" in n]
t0 = knn[0][0]
prev = None
for s, e in knn:
    print("knn %8.3f .. %8.3f  (%.2f ms)  gap %+.3f ms" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, (s - prev) / 1e6 if prev else 0.0))
    prev = e
# coverage between the start of the 2nd and the end of the (n-3)rd launch (skips warm-up and the single-batch tail)
lo, hi = knn[1][0], knn[-4][1] if len(knn) > 6 else knn[-1][1]
ev = sorted([(s, 1) for s, e in knn] + [(e, -1) for s, e in knn])
cov = {0: 0, 1: 0, 2: 0}
depth, last = 0, None
for t, d in ev:
    if last is not None and t > lo and last < hi:
        a, b = max(last, lo), min(t, hi)
        if b > a: cov[min(depth, 2)] += b - a
    depth += d; last = t
tot = sum(cov.values()) or 1
print("window %.1f ms: no kNN resident %.1f %%, one %.1f %%, two or more %.1f %%" % (tot / 1e6, 100 * cov[0] / tot, 100 * cov[1] / tot, 100 * cov[2] / tot))
other = {}
for n, s, e in rows:
    if "knn_tile" in n or s < lo or e > hi: continue
    k = re.sub(r"\(.*", "", n).replace("slideo::", "").replace("void ", "")
    other[k] = other.get(k, 0) + (e - s)
print("other kernels inside the window (sum of durations, ms):", ", ".join("%s %.1f" % (k, v / 1e6) for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:8]))
