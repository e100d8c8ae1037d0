"""GPU tier: the HIP engine (through the C ABI) against the committed golden vectors -- no oracle
involved at run time."""
import numpy as np
import pytest

from golden_util import SCENARIOS, load
from util import KINDS, _PARAM_MAP, to_tx_records

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", SCENARIOS)
def test_engine_matches_golden(engine, rsa, name):
    g = load(name)
    engine.upload_table(g["nodes"])
    engine.set_model(KINDS[g["kind"]], **{_PARAM_MAP[k]: v for k, v in g["params"].items()})
    if g["matrix"] is not None:
        engine.set_n2n_matrix(g["matrix"])
    if g["seed"] is not None:
        engine.seed(g["seed"])
    for t in g["ticks"]:
        engine.tick_begin(t["begin"], t["begin"] + 1000)
        engine.enqueue_records(to_tx_records(rsa, t["packets"]))
        r = engine.tick_flush()
        assert r.count == len(t["pkt"])
        np.testing.assert_array_equal(r.pkt, t["pkt"])
        np.testing.assert_array_equal(r.dst, t["dst"])
        np.testing.assert_array_equal(r.verdict, t["verdict"])          # bit-exact decisions
        np.testing.assert_allclose(r.rssi, t["rssi"], rtol=1e-5, atol=0)  # north-star tolerance ...
        np.testing.assert_array_equal(r.rssi, t["rssi"])                  # ... and in fact bit-exact
        np.testing.assert_array_equal(r.sinr, t["sinr"])
        np.testing.assert_array_equal(r.pkt_interference, t["interference"])
    if g["seed"] is not None:
        assert engine.rng_state == g["final_rng_state"]


@pytest.mark.parametrize("name", [s for s in SCENARIOS if s != "logdist_sinr_overlap"])
def test_engine_matches_golden_through_one_batch(engine, rsa, name):
    """The same vectors with all ticks of a scenario in ONE rm_batch_run_device call (the media without an
    on-air list evaluate every packet on its own, so the ticks of a fixture are independent but for the
    generator, which the batch consumes tick by tick)."""
    from util import DeviceArray
    g = load(name)
    engine.upload_table(g["nodes"])
    engine.set_model(KINDS[g["kind"]], **{_PARAM_MAP[k]: v for k, v in g["params"].items()})
    if g["matrix"] is not None:
        engine.set_n2n_matrix(g["matrix"])
    if g["seed"] is not None:
        engine.seed(g["seed"])
    recs = [to_tx_records(rsa, t["packets"]) for t in g["ticks"]]
    dev = [DeviceArray(r) if len(r) else DeviceArray(nbytes=64) for r in recs]
    begins = [t["begin"] for t in g["ticks"]]
    engine.batch_run_device(begins, [b + 1000 for b in begins], [d.ptr.value for d in dev], [len(r) for r in recs])
    for b, t in enumerate(g["ticks"]):
        r = engine.batch_result_copy(b, len(recs[b]))
        assert r.count == len(t["pkt"])
        np.testing.assert_array_equal(r.pkt, t["pkt"])
        np.testing.assert_array_equal(r.dst, t["dst"])
        np.testing.assert_array_equal(r.verdict, t["verdict"])
        np.testing.assert_array_equal(r.rssi, t["rssi"])
        np.testing.assert_array_equal(r.pkt_interference, t["interference"])
    if g["seed"] is not None:
        assert engine.rng_state == g["final_rng_state"]
    for d in dev:
        d.free()
