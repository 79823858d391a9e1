"""bench.py output contract: one JSON line from rank 0 with the fields the driver reads, at N=1 and — control flow
only, both ranks on one GPU over gloo — at N=2."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    for k in REQUIRED + ["cpu_baseline"]:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["value"] > 0 and j["scaling"] == "weak" and j["data"] == "synthetic" and "workload" in j["config"]
    rf = j["roofline"]
    for k in ["bound", "achieved", "peak", "unit", "frac", "traffic"]:
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["frac"] <= 1        # the EXECUTED rate (K x Mu pairs)
    assert rf["algorithmic"]["frac"] >= rf["frac"]                                            # SURVEY 8(d)'s K x M count, reported beside it
    # the line's own cross-check: the top-level figure is anchored on the driver's clock —
    # frac x peak x ms_per_step == the flops a step's launches execute (so the kernel's time per step cannot exceed the step)
    flops = rf["frac"] * rf["peak"] * 1e12 * j["ms_per_step"] * 1e-3
    assert abs(flops - rf["flops_per_step"]) <= 0.02 * rf["flops_per_step"], (flops, rf["flops_per_step"])
    assert abs(rf["flops_per_step"] - 512.0 * rf["pairs_per_launch"] * rf["launches_per_step"]) <= 1e-6 * rf["flops_per_step"]
    pl = rf["per_launch"]                                                                      # the occupancy figure lives in a sub-record
    assert pl["avg_launch_ms"] > 0 and pl["launches"] >= 3 and 0 < pl["frac"] <= 1
    assert rf["shader_clock_mhz"] is None or 500 < rf["shader_clock_mhz"] < 2600              # measured by the search blocks themselves (ABI 7)
    hf = j["host_frames"]                                                                      # the PCIe-inclusive rate of the drop-in path, beside `value`
    assert hf["value"] > 0 and hf["h2d_inclusive_GBps"] > 0 and hf["verdicts_equal_to_resident_run"] is True
    assert rf["traffic"] is None or isinstance(rf["traffic"], (int, float))      # HBM bytes per launch; measured for the headline only
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["verdict_agreement_with_gpu"] == 1.0


def test_two_ranks_control_flow_over_gloo():
    env = dict(os.environ, SLIDEO_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=420, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    for k in REQUIRED:
        assert k in j, k
    assert j["n_gpus"] == 2 and j["steps"] == 2 and "cpu_baseline" not in j       # the CPU leg runs at N=1 only
    assert j["config"]["frames_per_step_per_gpu"] == 16 and j["value"] > 0
    assert j["config"]["collective"] == {"backend": "gloo", "all_gather_of_verdicts_checked": True}


def test_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` — no launcher, the form of the driver's recorded single-GPU command — starts the two ranks itself
    (torch.distributed.run on 127.0.0.1) and prints the one line of rank 0 (it used to exit with a hint)."""
    env = dict(os.environ, SLIDEO_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["value"] > 0
    assert j["config"]["collective"] == {"backend": "gloo", "all_gather_of_verdicts_checked": True}


def test_rccl_path_executes_on_one_gpu():
    """The driver's multi-GPU runs use the nccl (= RCCL) backend; on a single-GPU box that code — init_process_group("nccl"),
    the library writing the verdict records into the gather's DEVICE input, all_gather_into_tensor on the device, the
    max-over-ranks all_reduce — is executed at world size 1 under torchrun (SLIDEO_BENCH_FORCE_DIST=1)."""
    env = dict(os.environ, SLIDEO_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SLIDEO_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "tiny", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=420, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 1 and j["value"] > 0
    assert j["config"]["collective"] == {"backend": "nccl", "all_gather_of_verdicts_checked": True}


def test_strong_scaling_mode():
    """--total-frames: a fixed job (BASELINE configs[3] is one) split over steps and GPUs; the line says "strong"."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "4", "--warmup", "1",
                        "--total-frames", "96", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["scaling"] == "strong" and j["config"]["frames_per_step_per_gpu"] == 24 and j["value"] > 0


def test_cfg2_sift_l2_line():
    """--workload cfg2 (BASELINE configs[2]) as a complete matcher at a small size: SIFT on the device, the L2 2-NN on the int8
    matrix cores, ratio test, the path's own verify stages; verdicts against the truth, frame 0's SIFT output bit-exact against
    the CPU restatement and sampled neighbours against numpy, inside the run."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2", "--batch", "8", "--pages", "20",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["roofline"]["bound"] == "mfma" and j["roofline"]["kernel"] == "knn_l2_kernel" and j["sift_stage"]["bound"] == "hbm"
    ck = j["config"]["checked"]
    assert ck["knn_vs_numpy_64_queries"] is True and ck["sift_frame0_bit_exact_vs_cpu_restatement"] is True
    assert j["config"]["sift_vote"] == "tolerance" and j["config"]["accuracy_vs_synthetic_truth"] >= 0.85 and j["value"] > 0 and 0 < j["roofline"]["frac"] < 1
    assert set(j["stage_ms_per_batch"]) == {"sift", "l2_knn", "verify"}
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] == 1


def test_cfg2_full_size():
    """BASELINE configs[2] at its full size: 256 1080p frames x SIFT-1000 against the SIFT descriptors of 500 pages."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    c = j["config"]
    assert c["frames_per_step_per_gpu"] == 256 and c["pages"] == 500 and c["train_descriptors_M"] > 400000 and c["query_descriptors_per_step"] > 200000
    assert c["checked"]["knn_vs_numpy_64_queries"] is True and c["checked"]["sift_frame0_bit_exact_vs_cpu_restatement"] is True
    assert c["sift_vote"] == "tolerance" and c["accuracy_vs_synthetic_truth"] >= 0.97 and j["roofline"]["frac"] > 0.15     # (the path's own vote: 256 of 256; Lowe's test — --sift-vote ratio — reaches 0.71)


def test_cfg3_lecture_two_ranks_over_gloo():
    """--workload cfg3 = BASELINE configs[3] as worded (1080p lecture, 1000-page deck, a FIXED 216 000-frame job sharded over the
    GPUs, one all-gather of the verdict records per step, the page timeline on rank 0), here at a small size: 20 pages, 192 frames
    in all, two ranks sharing the GPU over gloo, a step's 48 frames per rank as three units of 16 cycling through the resident
    pool.  The gathered slice of every rank equals its own verdicts, and rank 0's timeline equals the truth's."""
    env = dict(os.environ, SLIDEO_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg3", "--pages", "20",
                        "--total-frames", "192", "--pool", "16", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    c = j["config"]
    assert j["scaling"] == "strong" and j["n_gpus"] == 2 and c["frames_per_step_per_gpu"] == 48 and c["units_per_step"] == 3
    assert c["resident_pool_frames_per_gpu"] == 16 and c["collective"] == {"backend": "gloo", "all_gather_of_verdicts_checked": True}
    lec = c["lecture"]
    assert lec["sampled_frames_in_the_last_step"] == 96 and lec["timeline_entries"] >= 2
    assert c["accuracy_vs_synthetic_truth"] >= 0.9 and lec["entries_equal_to_truth"] >= 0.8 * lec["truth_entries"]
    assert j["value"] > 0 and j["roofline"]["frac"] > 0 and j["roofline"]["launches_per_step"] == 3


def test_cfg3_eight_ranks_dry_run_and_one_rank_under_the_launcher():
    """Dry run of the driver's 8-GPU form of configs[3] on whatever this box has: `bench.py --gpus 8 --workload cfg3
    --backend gloo --share-device` starts its own eight ranks (they share the device; the verdict all-gather goes through host memory),
    the job is FIXED (strong scaling), every rank's gathered slice equals its own verdicts.  And the N = 1 line under the launcher
    (the RCCL path at world size 1) measures what the plain command measures: same job, same per-step work, rates within 10 %
    (20-page deck: a small job; the full-size comparison is in profiles/)."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SLIDEO_BENCH_BACKEND"):
        env.pop(k, None)
    common = ["--workload", "cfg3", "--pages", "20", "--pool", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-frames"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--share-device", "--total-frames", "512"] + common,
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    c = j["config"]
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and c["frames_per_step_per_gpu"] == 32 and c["units_per_step"] == 2
    assert c["collective"] == {"backend": "gloo", "all_gather_of_verdicts_checked": True}
    assert c["lecture"]["sampled_frames_in_the_last_step"] == 256 and j["value"] > 0
    # N = 1: plain command against the same line under torch.distributed.run with the process group forced (nccl at world size 1)
    # (units of 128 frames, two per step: the unit size of the full job — with the dry run's 16-frame units the per-unit host work of the
    # collective path, not the GPU, would be what is compared)
    one = ["--gpus", "1", "--workload", "cfg3", "--pages", "20", "--pool", "128", "--total-frames", "1536", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--no-host-frames"]
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + one, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stderr[-3000:]
    env2 = dict(env, SLIDEO_BENCH_FORCE_DIST="1", SLIDEO_BENCH_FORCE_LAUNCH="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + one, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env2)
    assert r2.returncode == 0, r2.stderr[-3000:]
    a, b = _json_line(r1.stdout), _json_line(r2.stdout)
    assert a["scaling"] == b["scaling"] == "strong" and a["config"]["frames_per_step_per_gpu"] == b["config"]["frames_per_step_per_gpu"] == 256
    assert a["config"]["collective"] is None and b["config"]["collective"] == {"backend": "nccl", "all_gather_of_verdicts_checked": True}
    assert abs(a["value"] - b["value"]) <= 0.10 * a["value"], (a["value"], b["value"])


def test_nccl_with_fewer_gpus_than_ranks_fails_fast():
    """Two nccl ranks on a one-GPU box: RCCL would hang in its first collective; bench.py says what is wrong and exits."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has a GPU per rank")
    env = dict(os.environ)
    env.pop("SLIDEO_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "needs one GPU per rank" in r.stderr, r.stderr[-2000:]
