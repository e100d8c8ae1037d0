"""Receiver-sharded evaluation on the GPU: two contexts with the two halves of the receivers see
the same gathered Tx slots (padding included); their merged heard links equal the global oracle
run.  (The all-gather itself is covered on CPU by tests/test_dist_gloo.py.)"""
import numpy as np
import pytest

from util import to_tx_records, KINDS, _PARAM_MAP, oracle_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,params", [("udgm", {}), ("logdist", {"ld_sigma_db": 4.0, "ld_seed": 5}),
                                         ("logdist", {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 6})])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_global(rsa, O, kind, params, world):
    from radio_sim_amd import dist as D
    n = 3001
    rng = np.random.default_rng(17)
    nd = O.NodeTable(n)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    nd.channel[:] = 11 + rng.integers(0, 3, n)
    srcs = np.sort(rng.choice(n, 90, replace=False)).astype(np.int32)
    slots = D.slots_needed(n, world, [srcs])
    # what the all-gather would deliver: per rank `slots` records, padded with src = -1
    parts = []
    for r in range(world):
        lo, hi = D.partition(n, r, world)
        mine = srcs[(srcs >= lo) & (srcs < hi)]
        pk = nd.packets(mine, 0, 8128)
        pk["start_us"] = rng.integers(0, 1000, len(pk))
        parts.append(D.pad_records(to_tx_records(rsa, pk), slots))
    gathered = np.concatenate(parts)
    valid, slot_idx = D.drop_padding(gathered)

    shards = []
    for r in range(world):
        lo, hi = D.partition(n, r, world)
        eng = rsa.Engine(0)
        try:
            eng.upload_table(nd)
            eng.set_model(KINDS[kind], **{_PARAM_MAP[k]: v for k, v in params.items()})
            eng.set_partition(lo, hi - lo)
            res = eng.tick(gathered)
            assert res.count == 0 or (res.dst.min() >= lo and res.dst.max() < hi)
            shards.append((res.pkt, res.dst, res.verdict, res.rssi, res.sinr))
        finally:
            eng.close()
    pkt, dst, verdict, rssi, sinr = D.merge_shard_links(shards, world * slots)

    pk = np.zeros(len(valid), dtype=O.PACKET_DTYPE)
    for f in ("src", "channel", "x", "y", "z", "txpower", "txprob", "start_us", "air_us"):
        pk[f] = valid[f]
    ref = O.tick(oracle_model(O, kind, params), nd, pk)
    assert ref.count > 300 and len(pkt) == ref.count
    np.testing.assert_array_equal(pkt, slot_idx[ref.pkt])
    np.testing.assert_array_equal(dst, ref.dst)
    np.testing.assert_array_equal(verdict, ref.verdict)
    np.testing.assert_array_equal(rssi, ref.rssi)
    np.testing.assert_array_equal(sinr, ref.sinr)


def test_partition_with_draws_is_refused(rsa, O):
    """Probabilistic links need the global draw order: refused loudly, never silently wrong."""
    nd = O.NodeTable(200)
    nd.x = np.arange(200.0)
    nd.rxprob[:] = 0.5
    eng = rsa.Engine(0)
    try:
        eng.upload_table(nd)
        eng.set_model(rsa.MODEL_UDGM)
        eng.set_partition(0, 100)
        with pytest.raises(rsa.RadioMediumError) as e:
            eng.tick(to_tx_records(rsa, nd.packets([5])))
        assert e.value.code == -5
    finally:
        eng.close()


def test_pipelined_sharded_driver_on_one_gpu():
    """The two-stream tick driver of the multi-GPU path (pack + gather of tick t+1 overlapping the
    sweep of tick t), run with world == 1: it must see exactly the heard links the plain path sees.
    Run in fresh processes: the driver needs torch, and torch's bundled HIP runtime must be the
    first one loaded in its process."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra in ([], ["--force-sharded"]):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "c2", "--steps", "40",
                            "--warmup", "5", "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
        outs.append(json.loads(line))
    assert outs[0]["config"]["heard_links_last_tick"] == outs[1]["config"]["heard_links_last_tick"] > 0
    assert outs[1]["value"] > 0
