"""bench.py's launcher logic without a GPU: `python bench.py --gpus N` started with no launcher re-executes itself under
torch.distributed.run with one rank per GPU (the driver's 8-GPU command), and every rank then stops at the first thing it needs --
a GPU -- with bench.py's own message, not with the old "launch with torch.distributed.run" exit."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="on a GPU box the command would run the whole N-rank bench")
def test_bench_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert out.count("bench.py needs a GPU") >= 2, out[-2000:]            # both ranks came up under the launcher
    assert "launch with torch.distributed.run" not in out


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check of the argument handling")
def test_world_size_must_match_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)
