"""The collective path on REAL several GPUs -- armed when the box has two or more (skipped, not failed, on one): fresh child
processes, one per device (nothing in them has touched a GPU before they pick theirs), join a communicator with
rm_comm_init_rank and run rm_dist_batch_run_sources_device (incl. a batch of SINR ticks whose frames outlive their tick) and
rm_dist_tick_run_sources_device (java.util.Random draws; SINR with frames on the air) over RCCL / xGMI; the ranks' links,
merged by node index, have to be the one-process oracle's.  rm_group_tick_run_sources_device over ncclCommInitAll with one
member per device, and `bench.py --gpus N` over the real backend, the same way.  On the one-GPU box of a round these tests
skip; the world-2 gloo test (tests/test_dist_gloo.py) and the several-contexts-on-one-GPU tests (test_gpu_sharded.py,
test_gpu_group.py, test_gpu_comm.py, test_gpu_overlap.py) cover the same code minus the wire."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices():
    from radio_sim_amd import _lib
    return int(_lib.lib().rm_device_count())


def _need(n=2):
    d = _devices()
    if d < n:
        pytest.skip("needs %d GPUs, the box has %d: the collective path over several devices is armed for the day it has" % (n, d))
    return min(d, 8)


def _child_env():
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "RM_FORCE_DEVICE", "RM_DIST_BACKEND"):
        e.pop(k, None)
    return e


def _merge(parts, renum):
    """the ranks' links of one tick, packet-major and node index ascending inside a packet"""
    pk = np.concatenate([renum[p["pkt"]] for p in parts])
    dst = np.concatenate([p["dst"] for p in parts])
    key = np.lexsort((dst, pk))
    out = {f: np.concatenate([p[f] for p in parts])[key] for f in ("dst", "verdict", "rssi", "sinr")}
    out["pkt"] = pk[key]
    return out


def test_ranks_over_rccl_equal_the_oracle(O):
    _ranks_against_the_oracle(O, _need(2))


def test_rank_worker_rehearsal_with_one_rank(O):
    """the same worker, scenarios and merge with a world of ONE (RCCL admits one rank per device): what a one-GPU box can run of it"""
    _ranks_against_the_oracle(O, 1)


def _ranks_against_the_oracle(O, world):
    sys.path.insert(0, os.path.join(ROOT, "tests", "multi"))
    from rank_worker import scenario_inputs
    n = 24_000
    with tempfile.TemporaryDirectory() as d:
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multi", "rank_worker.py"), d, str(r), str(world)],
                                  env=_child_env(), stderr=subprocess.PIPE, text=True) for r in range(world)]
        errs = [p.communicate(timeout=900)[1] for p in procs]
        for r, p in enumerate(procs):
            assert p.returncode == 0, "rank %d: %s" % (r, errs[r][-3000:])
        ranks = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(world)]
    x, y, rxprob, ticks = scenario_inputs(n, world)
    own = ranks[0]["own"]
    nd = O.NodeTable(n)
    nd.x, nd.y = x, y

    def check(name, mdl, starts, air, chain_air=False, seed=None):
        slots = int(ranks[0][name + "_slots"][0])
        onair = np.zeros(0, dtype=O.PACKET_DTYPE)
        state = O.lib().orc_jrandom_seed(seed) if seed is not None else 0
        for b, srcs in enumerate(ticks[name]):
            order = np.concatenate([np.concatenate([srcs[own[srcs] == r], np.full(slots, -1, np.int32)])[:slots] for r in range(world)])
            real = order >= 0
            renum = np.cumsum(real) - 1
            new = nd.packets(order[real], starts[b], air)
            if chain_air:
                onair = onair[onair["start_us"] + onair["air_us"] > starts[b]]
            active = np.concatenate([onair, new]) if chain_air else new
            cpu = O.tick(mdl, nd, active, first_new=len(active) - len(new), rng_state=state, cap=1 << 22)
            state = cpu.rng_state
            if chain_air:
                onair = active
            parts = [{f: ranks[r]["%s_%d_%s" % (name, b, f)] for f in ("pkt", "dst", "verdict", "rssi", "sinr")} for r in range(world)]
            got = _merge(parts, renum)
            assert len(got["pkt"]) == cpu.count > 1000, (name, b, len(got["pkt"]), cpu.count)
            np.testing.assert_array_equal(got["pkt"], cpu.pkt, err_msg="%s tick %d" % (name, b))
            np.testing.assert_array_equal(got["dst"], cpu.dst, err_msg="%s tick %d" % (name, b))
            np.testing.assert_array_equal(got["verdict"], cpu.verdict, err_msg="%s tick %d" % (name, b))
            np.testing.assert_array_equal(got["rssi"], cpu.rssi, err_msg="%s tick %d" % (name, b))
            if chain_air:
                np.testing.assert_array_equal(got["sinr"], cpu.sinr, err_msg="%s tick %d" % (name, b))
            if seed is not None:
                for r in range(world):     # every rank ends the tick with the same generator state: the one-process one
                    assert int(ranks[r]["%s_%d_rng" % (name, b)][0]) == cpu.rng_state

    ld = {"ld_sigma_db": 4.0, "ld_seed": 9}
    check("batch", O.model(O.MODEL_LOGDIST, **ld), [b * 1000 for b in range(6)], 8128)
    check("overlap", O.model(O.MODEL_LOGDIST, ld_flags=1, **ld), [(100 + b) * 1000 for b in range(10)], 8128, chain_air=True)
    nd.rxprob = rxprob
    check("draws", O.model(O.MODEL_UDGM, udgm_success_ratio_rx=0.9), [300_000 + b * 1000 for b in range(3)], 8128, seed=77)
    nd.rxprob = np.ones(n)
    check("sinr_tick", O.model(O.MODEL_LOGDIST, ld_flags=1, **ld), [500_000 + b * 1000 for b in range(4)], 8128, chain_air=True)


def test_group_with_one_member_per_device(O):
    _group(_need(2))


def test_group_worker_rehearsal_with_one_device(O):
    _group(1)


def _group(world):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multi", "group_worker.py"), str(world)], env=_child_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "identical to the oracle" in p.stdout


@pytest.mark.parametrize("workload", ["c2", "c4", "c5"])
def test_bench_over_the_real_backend(workload):
    """`python bench.py --gpus N` as the driver starts it, over RCCL: the ranks together hear what one GPU hears"""
    _need(2)
    common = ["--workload", workload, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-weak-probe", "--no-host-transfer",
              "--no-scale-probe", "--batch", "8"]

    def run(args):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_child_env(), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[:1000]
        return json.loads(lines[0])

    one = run(common)
    two = run(["--gpus", "2"] + common)
    assert two["n_gpus"] == 2 and "ncclAllGather inside libradiomedium_hip.so" in two["config"]["sharding"]
    assert two["config"]["heard_links_last_tick"] == one["config"]["heard_links_last_tick"] > 0
    assert two["value"] > 0
