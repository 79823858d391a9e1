cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tl2
rocprofv3 --kernel-trace -d gpurun_out/tl2 -o t -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/tl2/bench.log 2>&1
python - <<'PY'
import sqlite3
c = sqlite3.connect("gpurun_out/tl2/t_results.db")
print([r[1] for r in c.execute("pragma table_info(kernels)")])
rows = c.execute("select * from kernels order by start limit 2").fetchall()
print(rows)
import re
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
print("qcol", qcol)
sel = "select name, start, duration%s from kernels order by start" % ((", " + qcol) if qcol else "")
ev = c.execute(sel).fetchall()
t0 = ev[0][1]
# per queue: the sequence of stage intervals
def grp(n):
    if "knn_tile" in n and "expand" not in n: return "K"
    if any(k in n for k in ("gray_kernel","resize_kernel","fast_kernel","threshold_kernel","scan_kernel","compact_kernel","sort_kernel","blur_","describe")): return "o"
    if any(k in n for k in ("vote_kernel","ransac","rate_kernel","reproject","verdict","expand_dups")): return "v"
    return None
from collections import defaultdict
per = defaultdict(list)
for e in ev:
    g = grp(e[0]); q = e[3] if qcol else 0
    if not g: continue
    L = per[q]
    if L and L[-1][0] == g and e[1] - L[-1][2] < 3e6: L[-1][2] = e[1] + e[2]
    else: L.append([g, e[1], e[1] + e[2]])
for q, L in per.items():
    print("queue", q)
    for g, s, e in L[-30:]:
        print("   %s  %9.3f -> %9.3f  (%6.3f ms)" % (g, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
PY
rm -f gpurun_out/tl2/t_results.db
