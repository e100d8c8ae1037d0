"""Several contexts behind one caller (rm_group_*, SURVEY.md section 8e for a single-process host): receivers
partitioned over the members -- by region of the plane (the default) or by node index range --, a tick's Tx records handed
to every member from the host (no all-gather), the per-packet java.util.Random draw counts (regions: also the drawing
links' nodes) exchanged through the host, the members' links merged by node index.  On a one-GPU box every member sits on device 0;
the result must be what one context -- and the oracle -- gives: links, order, verdicts, Tx-failure flags and the
generator state after every tick."""
import numpy as np
import pytest

from util import KINDS, _PARAM_MAP, assert_same, oracle_model, random_nodes, to_tx_records

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("members,spatial", [(2, False), (3, False), (2, True), (3, True), (8, True)])
@pytest.mark.parametrize("kind,params", [("udgm", {}), ("udgm", {"udgm_success_ratio_rx": 0.8}),
                                         ("logdist", {"ld_sigma_db": 4.0, "ld_seed": 5}), ("null", {}), ("udgm_const", {})])
def test_group_equals_one_context_and_the_oracle(rsa, O, members, spatial, kind, params):
    n = 3000 if kind != "null" else 400
    rng = np.random.default_rng(17)
    nd = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), seed=17)
    if params.get("udgm_success_ratio_rx", 1.0) != 1.0:
        nd.rxprob[rng.choice(n, n // 4, replace=False)] = 0.5
        nd.txprob[rng.choice(n, n // 10, replace=False)] = 0.7
    g = rsa.Group([0] * members, spatial=spatial)
    one = rsa.Engine(0)
    try:
        kw = {_PARAM_MAP[k]: v for k, v in params.items()}
        for e in (g, one):
            e.upload_table(nd)
            e.set_model(KINDS[kind], **kw)
            e.seed(99)
        assert g.size == members
        mdl = oracle_model(O, kind, params)
        state = O.lib().orc_jrandom_seed(99)
        for k in range(4):
            t = int(rng.integers(1, 70)) if k else 64
            pk = nd.packets(rng.choice(n, t, replace=False), k * 1000, 8128)
            cpu = O.tick(mdl, nd, pk, rng_state=state)
            state = cpu.rng_state
            got = g.tick(to_tx_records(rsa, pk), k * 1000, k * 1000 + 1000)
            ref = one.tick(to_tx_records(rsa, pk), k * 1000, k * 1000 + 1000)
            assert_same(got, cpu, "%s group of %d, tick %d vs oracle" % (kind, members, k))
            assert_same(got, ref, "%s group of %d, tick %d vs one context" % (kind, members, k))
            np.testing.assert_array_equal(got.pkt_offset, ref.pkt_offset)
            assert g.rng_state == one.rng_state == cpu.rng_state
        # an empty tick
        empty = g.tick(np.zeros(0, dtype=rsa.TX_RECORD_DTYPE), 5000, 6000)
        assert empty.count == 0
    finally:
        g.close()
        one.close()


def test_group_c_abi_from_a_plain_loop(rsa, O):
    """the group through its C entry points, the way a single-threaded host drives it: rm_group_enqueue_tx per packet"""
    import ctypes as C
    from radio_sim_amd import _lib
    L = _lib.lib()
    n = 1000
    nd = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), seed=3)
    g = rsa.Group([0, 0])
    try:
        g.upload_table(nd)
        g.set_model(KINDS["udgm"])
        assert L.rm_group_context(g._h, 0) and L.rm_group_context(g._h, 1) and not L.rm_group_context(g._h, 2)
        assert L.rm_group_tick_begin(g._h, 0, 1000) == 0
        srcs = [5, 500, 999]
        for s in srcs:
            assert L.rm_group_enqueue_tx(g._h, s, 10, 320, None, None) == 0
        cap = 4096
        dst = np.empty(cap, dtype=np.int32)
        pkt = np.empty(cap, dtype=np.int32)
        verdict = np.empty(cap, dtype=np.uint8)
        rssi = np.empty(cap, dtype=np.float64)
        poff = np.empty(4, dtype=np.uint32)
        cnt = C.c_uint32(0)
        assert L.rm_group_tick_flush(g._h, pkt.ctypes.data, dst.ctypes.data, verdict.ctypes.data, rssi.ctypes.data, None, cap,
                                     C.byref(cnt), None, poff.ctypes.data) == 0
        cpu = O.tick(oracle_model(O, "udgm", {}), nd, nd.packets(srcs, 10, 320))
        assert cnt.value == cpu.count == poff[3]
        np.testing.assert_array_equal(dst[:cnt.value], cpu.dst)
        np.testing.assert_array_equal(pkt[:cnt.value], cpu.pkt)
    finally:
        g.close()
