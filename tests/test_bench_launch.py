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


# ---- the roofline line's arithmetic, on canned kernel tables (no device: bench.py imports numpy only at module level)

def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _table(**us):   # kernel -> (launches, total ms): 24 sampled sequences, one launch of each kernel per sequence
    return {k: {"launches": 24, "ms": v * 24 / 1e3, "stage": 0} for k, v in us.items()}


def test_roofline_object_follows_its_definition():
    """frac = required bytes of a launch / the dominant kernel's interval ALONE on the device / 8 TB/s; the contended intervals of
    the timed region only feed the cross-check; `bound` is the resource the step uses most of; nothing above 1 is printed."""
    import pytest
    B = _bench()
    n_loc, t, heard, cand, ticks = 100_000, 1000, 44_000, 46_000, 128
    contended = _table(**{"k_tick_prep_batch": 30.0, "k_filter_wg_batch<4, true>": 390.0, "k_exact_batch<RM_MODEL_LOGDIST, false, 3>": 415.0,
                          "k_reorder_batch<false, 3>": 208.0})
    alone = {"kernels": _table(**{"k_tick_prep_batch": 6.0, "k_filter_wg_batch<4, true>": 170.0, "k_exact_batch<RM_MODEL_LOGDIST, false, 3>": 130.0,
                                  "k_reorder_batch<false, 3>": 100.0}),
             "n_samples": 24, "sequences": 24, "step_us_plain": 415.0, "step_us_probed": 421.0, "launches_per_sequence": 4.0,
             "probe_cost_us_per_launch": 1.5}
    rl = B.roofline_object(kernels=contended, n_samples=24, n_loc=n_loc, t_per_tick=t, heard=heard, cand=cand, ticks_per_launch=ticks,
                           step_s=370e-6, contexts=3, workload="no-such-workload", pmc_ok=False, alone=alone, sinr_column=False, tile_reuse=4)
    req = int((n_loc * 37 / 4 + t * 56 + heard * 17) * ticks)
    assert rl["required_bytes_per_launch"] == req == B.required_bytes(n_loc, t, heard, ticks, False, 4)
    assert rl["algorithmic_bytes_per_launch"] == (n_loc * 37 + t * 56 + heard * 25) * ticks
    assert rl["kernel"] == "k_filter_wg_batch<4, true>" and rl["kernel_avg_us"] == pytest.approx(170.0)   # dominant ALONE, not contended
    assert rl["achieved"] == pytest.approx(req / 170e-6 / 1e9)
    assert rl["frac"] == pytest.approx(rl["achieved"] / 8000.0) and rl["frac"] <= 1.0
    assert rl["frac_8d"] == pytest.approx((n_loc * 37 + t * 56 + heard * 25) * ticks / 170e-6 / 1e9 / 8000.0)
    oc = rl["overlap_check"]
    assert oc["per_context_share_us"] == pytest.approx((30 + 390 + 415 + 208) / 3.0)
    assert oc["probe_slack_us"] == pytest.approx(1.5 * 4 / 3.0) and oc["ok"] is True      # 347.7 <= 370 * 1.02 + 2
    assert rl["alone"]["ok"] is True and rl["alone"]["kernel_us_per_sequence"] == pytest.approx(406.0)
    assert rl["bound"] == "hbm" and rl["step"]["valu_frac"] is None                        # no counter file for this name
    # a step shorter than its kernels' per-context share: flagged, the fraction unaffected
    rl2 = B.roofline_object(kernels=contended, n_samples=24, n_loc=n_loc, t_per_tick=t, heard=heard, cand=cand, ticks_per_launch=ticks,
                            step_s=300e-6, contexts=3, workload="no-such-workload", pmc_ok=False, alone=alone, tile_reuse=4, under_profiler=False)
    assert rl2["overlap_check"]["ok"] is False and rl2["frac"] == rl["frac"]
    # the SINR medium writes 25-byte records; a lone tick charges the table every tick
    assert B.required_bytes(1000, 10, 100, 1, True) == 1000 * 37 + 10 * 56 + 100 * 25
    # a kernel cannot have moved its launch's bytes faster than the HBM: such a line is refused
    fast = dict(alone, kernels=_table(**{"k_filter_wg_batch<4, true>": 5.0}))
    with pytest.raises(SystemExit):
        B.roofline_object(kernels=contended, n_samples=24, n_loc=n_loc, t_per_tick=t, heard=heard, cand=cand, ticks_per_launch=ticks,
                          step_s=370e-6, contexts=3, workload="no-such-workload", pmc_ok=False, alone=fast, tile_reuse=1)


def test_roofline_bound_is_the_larger_fraction():
    """with the counter passes on file for configs[2] the step's vector-issue share (0.6 - 0.7) exceeds its HBM share (0.2): valu"""
    B = _bench()
    if not os.path.exists(os.path.join(ROOT, "profiles", "pmc_traffic.json")):
        return
    k = _table(**{"k_filter_wg_batch<4, true>": 170.0, "k_exact_batch<RM_MODEL_LOGDIST, false, 3>": 130.0, "k_reorder_batch<false, 3>": 100.0})
    rl = B.roofline_object(kernels=k, n_samples=24, n_loc=100_000, t_per_tick=1000, heard=44_000, cand=46_000, ticks_per_launch=128,
                           step_s=380e-6, contexts=1, workload="c3", alone={"kernels": k, "n_samples": 24, "probe_cost_us_per_launch": 1.0,
                                                                            "step_us_probed": 420.0}, tile_reuse=4)
    assert rl["valu_issue"] is not None and rl["step"]["valu_frac"] > rl["step"]["hbm_frac"] and rl["bound"] == "valu"
    assert rl["traffic"] is not None and rl["traffic_over_required"] > 0
