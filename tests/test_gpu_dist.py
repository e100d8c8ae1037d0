"""bench.py's multi-rank path as the driver starts it -- `python bench.py --gpus N`, no launcher -- rehearsed on the one GPU
of the test box: two ranks over gloo share device 0 (RM_DIST_BACKEND=gloo RM_FORCE_DEVICE=0; RCCL admits one rank per
device, so the collective is torch.distributed's here), receivers partitioned by region; and the library's own collective
(--collective lib: rm_comm_init_rank + rm_dist_batch_run_sources_device, RCCL inside libradiomedium_hip.so) with one rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, timeout=600):
    e = dict(os.environ, **(env or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[:1000]
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_on_one_gpu():
    common = ["--workload", "c2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-weak-probe", "--no-host-transfer",
              "--no-scale-probe", "--batch", "16"]
    one = _bench(common)
    for part in ("spatial", "index"):
        two = _bench(["--gpus", "2", "--partition", part] + common, env={"RM_DIST_BACKEND": "gloo", "RM_FORCE_DEVICE": "0"})
        assert two["n_gpus"] == 2 and two["steps"] == 3 and two["warmup"] == 1
        assert two["config"]["nodes"] == one["config"]["nodes"] and two["scaling"] == "strong"
        # the two ranks together hear what one context hears (the last tick of the run, summed over the ranks)
        assert two["config"]["heard_links_last_tick"] == one["config"]["heard_links_last_tick"] > 0
        assert two["value"] > 0


def test_library_collective_with_one_rank():
    common = ["--workload", "c2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-host-transfer", "--no-scale-probe",
              "--batch", "16"]
    plain = _bench(common)
    lib = _bench(common + ["--collective", "lib"])
    assert lib["config"]["heard_links_last_tick"] == plain["config"]["heard_links_last_tick"] > 0
    assert lib["config"]["ticks_per_launch"] == 16 and lib["value"] > 0
