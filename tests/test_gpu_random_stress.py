"""Seeded randomized parity sweep: small, awkward scenarios over all media and parameter corners
(the conservative pre-filter, the shadowing table, the on-air list and the draw machinery all have
special cases: thr = -1 / +inf, exponent 0, clip 0, d < d0, coincident nodes, ragged sizes ...)."""
import numpy as np
import pytest

from util import configure_engine, oracle_model, to_tx_records, assert_same

pytestmark = pytest.mark.gpu


def _scenario(O, rng):
    n = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 129, 257, 700, 1500]))
    kind = str(rng.choice(["null", "udgm", "udgm_const", "n2n", "logdist", "logdist", "logdist_sinr"]))
    nd = O.NodeTable(n)
    layout = rng.choice(["uniform", "line", "clustered", "coincident", "3d"])
    side = float(rng.choice([10.0, 200.0, 2000.0]))
    if layout == "uniform":
        nd.x, nd.y = rng.uniform(-side, side, n), rng.uniform(-side, side, n)
    elif layout == "line":
        nd.x = np.arange(n) * float(rng.choice([0.5, 10.0, 50.0]))
    elif layout == "clustered":
        c = rng.uniform(-side, side, (4, 2))
        k = rng.integers(0, 4, n)
        nd.x, nd.y = c[k, 0] + rng.normal(0, 15, n), c[k, 1] + rng.normal(0, 15, n)
    elif layout == "coincident":
        nd.x[:] = 5.0
        nd.y[:] = -3.0
    else:
        nd.x, nd.y, nd.z = rng.uniform(0, side, n), rng.uniform(0, side, n), rng.uniform(0, side / 4, n)
    nd.channel[:] = rng.choice([26, 26, 26, 11, -4, 2 ** 31 - 1], n)
    nd.enabled[:] = rng.random(n) > rng.choice([0.0, 0.1, 1.0, 0.5])
    nd.txpower[:] = rng.choice([0.0, -25.0, 20.0, -300.0, 5.5], n)
    if rng.random() < 0.5:
        nd.rxprob[:] = rng.choice([1.0, 0.0, 0.3, 1.5, -0.2], n)
        nd.txprob[:] = rng.choice([1.0, 0.0, 0.6, 2.0], n)
    params, matrix = {}, None
    if kind == "udgm":
        params = {"udgm_transmission_range": float(rng.choice([50.0, 0.0, 1e-3, 1e9, -50.0])),
                  "udgm_success_ratio_rx": float(rng.choice([1.0, 0.5, 0.0, 1.2]))}
    elif kind == "udgm_const":
        params = {"const_range": float(rng.choice([100.0, 0.0, -1.0, 1e12, 10.0]))}
    elif kind == "n2n":
        m = int(rng.choice([n, max(1, n // 2), n + 3]))
        matrix = np.where(rng.random((m, m)) < 0.3, rng.uniform(-0.1, 1.3, (m, m)), 0.0)
        nd.int_id[:] = rng.choice([1, 2, 3, -1, 0, m, m + 1], n) if rng.random() < 0.3 else np.arange(1, n + 1)
    else:
        params = {"ld_exponent": float(rng.choice([3.0, 2.0, 0.0, 6.5])), "ld_d0": float(rng.choice([1.0, 30.0, 0.01])),
                  "ld_pl0_db": float(rng.choice([40.0, 0.0, 90.0])), "ld_sigma_db": float(rng.choice([0.0, 4.0, 12.0])),
                  "ld_clip": float(rng.choice([3.0, 0.0, 1.0])), "ld_seed": int(rng.integers(0, 2 ** 40)),
                  "ld_sensitivity_dbm": float(rng.choice([-95.0, -40.0, -200.0]))}
        if kind == "logdist_sinr":
            params.update({"ld_flags": 1, "ld_capture_db": float(rng.choice([3.0, -10.0, 30.0])),
                           "ld_ifloor_dbm": float(rng.choice([-110.0, -60.0, -95.0, -300.0])),
                           "ld_noise_dbm": float(rng.choice([-100.0, -60.0]))})
        kind = "logdist"
    t = int(min(n, rng.choice([1, 2, 63, 64, 65, 130])))
    src = rng.choice(n, t, replace=False)
    pk = nd.packets(src, 0, 0)
    pk["start_us"] = rng.integers(0, 2000, t)
    pk["air_us"] = rng.choice([0, 320, 2000, 8128], t)
    if rng.random() < 0.3:
        pk["txpower"] = rng.uniform(-60, 30, t)
        pk["channel"] = rng.choice([26, 11], t)
    return nd, kind, params, matrix, pk


@pytest.mark.parametrize("block", range(int(__import__("os").environ.get("RM_STRESS_BLOCKS", "8"))))
def test_randomized_scenarios(engine, rsa, O, block):
    rng = np.random.default_rng(1000 + block)
    checked = 0
    for it in range(12):
        nd, kind, params, matrix, pk = _scenario(O, rng)
        if kind == "logdist" and params.get("ld_flags") and (len(pk) * nd.n > 60_000):
            pk = pk[:40]
        configure_engine(engine, nd, kind, params, matrix)
        seed = int(rng.integers(0, 2 ** 31))
        engine.seed(seed)
        cpu = O.tick(oracle_model(O, kind, params, matrix), nd, pk, rng_state=O.lib().orc_jrandom_seed(seed))
        gpu = engine.tick(to_tx_records(rsa, pk))
        assert_same(gpu, cpu, "block %d it %d %s %s n=%d t=%d" % (block, it, kind, params, nd.n, len(pk)))
        assert engine.rng_state == cpu.rng_state
        checked += cpu.count
    assert checked > 0


@pytest.mark.parametrize("block", range(int(__import__("os").environ.get("RM_STRESS_BLOCKS", "16"))))
def test_randomized_scenarios_through_batches_and_single_packets(engine, rsa, O, block):
    """The same awkward scenarios through the other two entry points: the tick's packets split over the
    ticks of an rm_batch_run_device call (batched kernels where they apply, one launch sequence per
    tick where they do not), and one rm_transmit per packet (k_transmit_one or its hand-over to the
    general path).  Without the SINR extension every packet is evaluated on its own, so both must give
    the oracle's per-packet answers and consume the generator exactly as the one-tick run does."""
    from util import DeviceArray
    rng = np.random.default_rng(5000 + block)
    checked = 0
    for it in range(10):
        nd, kind, params, matrix, pk = _scenario(O, rng)
        if params.get("ld_flags"):
            continue
        mdl = oracle_model(O, kind, params, matrix)
        seed = int(rng.integers(0, 2 ** 31))
        what = "block %d it %d %s %s n=%d t=%d" % (block, it, kind, params, nd.n, len(pk))
        # -- a batch: the packets cut into up to 5 ticks (some possibly empty)
        cuts = np.sort(rng.integers(0, len(pk) + 1, int(rng.integers(0, 5))))
        parts = np.split(pk, cuts)
        configure_engine(engine, nd, kind, params, matrix)
        engine.seed(seed)
        recs = [to_tx_records(rsa, p) for p in parts]
        dev = [DeviceArray(r) if len(r) else DeviceArray(nbytes=64) for r in recs]
        tb = [1000 * b for b in range(len(parts))]
        engine.batch_run_device(tb, [t + 1000 for t in tb], [d.ptr.value for d in dev], [len(r) for r in recs])
        state = O.lib().orc_jrandom_seed(seed)
        for b, p in enumerate(parts):
            cpu = O.tick(mdl, nd, p, rng_state=state)
            state = cpu.rng_state
            assert_same(engine.batch_result_copy(b, len(p)), cpu, what + " batch tick %d" % b)
            checked += cpu.count
        assert engine.rng_state == state
        for d in dev:
            d.free()
        # -- one call per packet (the packet's own rf-power / channel as overrides)
        engine.seed(seed)
        state = O.lib().orc_jrandom_seed(seed)
        for i in range(min(len(pk), 12)):
            one = pk[i:i + 1]
            hexlen = int(one["air_us"][0] // 32)
            got = engine.transmit(int(one["src"][0]), start_us=int(one["start_us"][0]), hex_length=hexlen,
                                  txpower=float(one["txpower"][0]), channel=int(one["channel"][0]))
            one = one.copy()
            one["air_us"] = hexlen * 32
            cpu = O.tick(mdl, nd, one, rng_state=state)
            state = cpu.rng_state
            assert got.count == cpu.count, what + " packet %d" % i
            np.testing.assert_array_equal(got.dst, cpu.dst, err_msg=what)
            np.testing.assert_array_equal(got.verdict, cpu.verdict, err_msg=what)
            np.testing.assert_array_equal(got.rssi, cpu.rssi, err_msg=what)
            assert bool(got.pkt_interference[0]) == bool(cpu.pkt_interference[0]), what
            assert engine.rng_state == state, what
    # (some blocks draw only scenarios without a single heard link: nothing to require of `checked` here)


@pytest.mark.parametrize("block", range(int(__import__("os").environ.get("RM_STRESS_BLOCKS", "8"))))
def test_randomized_scenarios_with_node_changes_between_ticks(engine, rsa, O, block):
    """node-config-set between ticks on the same awkward scenarios: single updates of every field and
    lists of moved nodes (a few metres, and teleports that make the table sort again), results read in
    place from the pinned block (rm_tick_flush_view) and through rm_batch_result_view."""
    from util import DeviceArray
    rng = np.random.default_rng(9000 + block)
    for it in range(8):
        nd, kind, params, matrix, pk = _scenario(O, rng)
        sinr = bool(params.get("ld_flags"))
        if sinr and (len(pk) * nd.n > 60_000):
            pk = pk[:40]
        pk["start_us"], pk["air_us"] = 0, 320          # self-contained ticks: nothing stays on the air
        configure_engine(engine, nd, kind, params, matrix)
        mdl = oracle_model(O, kind, params, matrix)
        seed = int(rng.integers(0, 2 ** 31))
        engine.seed(seed)
        state = O.lib().orc_jrandom_seed(seed)
        what = "block %d it %d %s %s n=%d t=%d" % (block, it, kind, params, nd.n, len(pk))
        n = nd.n
        for step in range(4):
            # single updates: any field
            for i in rng.choice(n, min(n, 3), replace=False):
                i = int(i)
                if rng.random() < 0.7:
                    nd.x[i] += rng.normal(0, 20.0)
                    nd.y[i] += rng.normal(0, 20.0)
                if rng.random() < 0.3:
                    nd.channel[i] = int(rng.choice([26, 11]))
                if rng.random() < 0.3:
                    nd.enabled[i] = int(rng.random() < 0.7)
                if rng.random() < 0.3:
                    nd.rxprob[i] = float(rng.choice([1.0, 0.4, 0.0]))
                    nd.txprob[i] = float(rng.choice([1.0, 0.7]))
                if rng.random() < 0.2:
                    nd.txpower[i] = float(rng.choice([0.0, -10.0, 7.0]))
                engine.update_node(i, nd.x[i], nd.y[i], nd.z[i], nd.txpower[i], int(nd.channel[i]), int(nd.enabled[i]),
                                   nd.rxprob[i], nd.txprob[i])
            # a list of moved nodes (z given or not)
            k = int(min(n, rng.choice([0, 1, 5, 70])))
            who = rng.choice(n, k, replace=False).astype(np.int32)
            if rng.random() < 0.3:
                nd.x[who], nd.y[who] = rng.uniform(-2000, 2000, k), rng.uniform(-2000, 2000, k)      # teleports
            else:
                nd.x[who] += rng.normal(0, 3.0, k)
                nd.y[who] += rng.normal(0, 3.0, k)
            if rng.random() < 0.5:
                nd.z[who] = 0.0
                engine.move_nodes(who, nd.x[who], nd.y[who])
            else:
                nd.z[who] = rng.uniform(0, 30.0, k)
                engine.move_nodes(who, nd.x[who], nd.y[who], nd.z[who])
            # the packets take the sources' current state
            cur = nd.packets(pk["src"], 1000 * step, 320)
            cpu = O.tick(mdl, nd, cur, rng_state=state)
            state = cpu.rng_state
            engine.tick_begin(1000 * step, 1000 * step + 1000)
            engine.enqueue_records(to_tx_records(rsa, cur))
            assert_same(engine.tick_flush_view(), cpu, what + " step %d" % step)
            assert engine.rng_state == state, what
        # the final state once more through a batch of two ticks and the batch view
        cur = nd.packets(pk["src"], 10000, 320)
        half = len(cur) // 2
        parts = [cur[:half], cur[half:]]
        parts[1]["start_us"] = 11000
        recs = [to_tx_records(rsa, p) for p in parts]
        dev = [DeviceArray(r) if len(r) else DeviceArray(nbytes=64) for r in recs]
        engine.batch_run_device([10000, 11000], [11000, 12000], [d.ptr.value for d in dev], [len(r) for r in recs])
        views, status = engine.batch_result_view(2)
        assert status == [0, 0], what
        for b in range(2):
            cpu = O.tick(mdl, nd, parts[b], rng_state=state)
            state = cpu.rng_state
            assert_same(views[b], cpu, what + " batch view %d" % b)
        assert engine.rng_state == state, what
        for d in dev:
            d.free()
