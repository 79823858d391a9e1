#!/usr/bin/env python3
"""profiles/knn_traffic.json from the PMC passes of tools/final_profile.sh.

usage: make_knn_traffic.py <final_dir> <tag>      (final_dir = gpurun_out/final_<tag>, copied to profiles/<tag>_pmc_hbm_traffic.txt)

bench.py reports the figure as roofline.traffic only while the kernel source (sha1 of csrc/knn_tile.hip.h) and the number of train
rows the kernel searched are the ones of the measured run — both are recorded here.
"""
import hashlib, json, os, re, sys

d, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def knn(path, counter):
    txt = open(path).read()
    m = re.search(r"slideo::knn_tile2_kernel: dispatches (\d+), total ([\d.]+) us\n\s+%s\s+(\d+)" % counter, txt)
    return int(m.group(1)), float(m.group(2)), int(m.group(3))


na, ta, fa = knn(os.path.join(d, "pmc_FETCH_SIZE.txt"), "FETCH_SIZE")
_, _, wa = knn(os.path.join(d, "pmc_WRITE_SIZE.txt"), "WRITE_SIZE")
no, to, fo = knn(os.path.join(d, "pmc_overlap_FETCH_SIZE.txt"), "FETCH_SIZE")
_, _, wo = knn(os.path.join(d, "pmc_overlap_WRITE_SIZE.txt"), "WRITE_SIZE")
line = json.loads([l for l in open(os.path.join(d, "bench_default.json")) if l.startswith("{")][-1])
mu = int(line["config"]["train_descriptors_unique"])
sha = hashlib.sha1(open(os.path.join(here, "slideo_amd", "csrc", "knn_tile.hip.h"), "rb").read()).hexdigest()
per = lambda kib, n: kib / n * 1024 / 1e9
out = {
    "_comment": "HBM-side bytes of knn_tile2_kernel (the default engine) on the headline workload, from separate rocprofv3 --pmc FETCH_SIZE / "
                "WRITE_SIZE passes (tools/final_profile.sh). bench.py reports them as roofline.traffic while kernel_source_sha1 and "
                "train_rows_searched match the run; FETCH_SIZE is doubled per the gfx950 correction.  `overlapped` = the same counters with four "
                "batches in flight (the timed configuration).",
    "source": "profiles/%s_pmc_hbm_traffic.txt" % tag,
    "kernel_source_sha1": sha,
    "train_rows_searched": mu,
    "dispatches": na, "fetch_size_kib": fa, "write_size_kib": wa,
    "overlapped": {"dispatches": no, "fetch_size_kib": fo, "write_size_kib": wo},
    "note": "per launch: %.2f GB fetched (x2-corrected; L2 -> fabric, served by the Infinity Cache that holds the whole FP4 matrix) + %.2f GB "
            "written alone; %.2f + %.2f GB with four batches in flight. The kernel searches the %d distinct rows of the train set "
            "(equal rows collapsed at finalize); its blocks stream that matrix unsynchronised, two per CU" % (
                per(2 * fa, na), per(wa, na), per(2 * fo, no), per(wo, no), mu),
}
json.dump(out, open(os.path.join(here, "profiles", "knn_traffic.json"), "w"), indent=2)
with open(os.path.join(here, "profiles", "%s_pmc_hbm_traffic.txt" % tag), "w") as f:
    for name in ("pmc_FETCH_SIZE.txt", "pmc_WRITE_SIZE.txt", "pmc_overlap_FETCH_SIZE.txt", "pmc_overlap_WRITE_SIZE.txt"):
        f.write("==== %s (%s) ====\n" % (name, "one batch in flight" if "overlap" not in name else "four batches in flight"))
        f.write(open(os.path.join(d, name)).read())
print(json.dumps(out, indent=1))
