"""`python bench.py --gpus N` must run as the driver runs it: without a launcher it starts its own N ranks as fresh child
processes (before anything touches a GPU), relays rank 0's one JSON line and the exit code.  CPU tier: the plumbing with
RM_BENCH_DRY_RUN=1 (gloo rendezvous, no device); the real two-rank run on one GPU is in tests/test_gpu_dist.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_spawns_its_own_ranks():
    env = dict(os.environ, RM_BENCH_DRY_RUN="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["rank_sum"] == 3.0          # both ranks took part in the collective


def test_parent_does_not_import_torch_before_spawning():
    """the parent must not have initialised anything GPU-related when it starts the ranks: torch is imported in the ranks"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[: src.index("def spawn_ranks")]
    assert "import torch" not in head
    body = src[src.index("def spawn_ranks"): src.index("def dry_run")]
    assert "import torch" not in body and "os.exec" not in body and "execv" not in body
