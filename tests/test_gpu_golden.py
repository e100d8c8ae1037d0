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
