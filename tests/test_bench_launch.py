"""`python bench.py --gpus N` without a launcher starts its own ranks (CPU-side check: the ranks come up and each says that the
bench needs a GPU — the N > 1 form used to stop with "launch with torch.distributed.run" before starting anything)."""
import os
import subprocess
import sys

import pytest

from conftest import HAS_GPU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(HAS_GPU, reason="with a GPU the same command runs the whole bench: tests/test_bench_contract.py")
def test_plain_gpus_2_spawns_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "[rank 0 of 2]" in r.stderr and "[rank 1 of 2]" in r.stderr, r.stderr[-2000:]
