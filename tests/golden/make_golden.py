#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

The reference (Java) cannot run here and ships no vectors of its own, so
these fixtures are outputs of oracle/rm_oracle.c -- which is itself pinned to the reference's
source by the known-answer tests of tests/test_oracle_kats.py.  They serve two purposes: the CPU
tier checks that the oracle still reproduces them bit for bit (compiler / platform drift), and the
GPU tier checks the HIP engine against them without needing the oracle at all.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
from util import KINDS  # noqa: E402

NODE_FIELDS = ("x", "y", "z", "txpower", "channel", "enabled", "rxprob", "txprob", "int_id")


def scenario(name):
    """-> (nodes, kind, params, matrix, seed, ticks) ; ticks = list of (t_begin, new packets)."""
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    if name == "udgm_default":
        n = 1500
        nd = O.NodeTable(n)
        side = 50.0 * np.sqrt(np.pi * n / 20.0)
        nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
        nd.channel[rng.random(n) < 0.1] = 25
        nd.enabled[rng.random(n) < 0.05] = 0
        nd.txpower[:] = rng.uniform(-25, 0, n)
        src = np.sort(rng.choice(n, 60, replace=False))
        return nd, "udgm", {}, None, None, [(0, nd.packets(src, 0, 8128))]
    if name == "udgm_stochastic":
        n = 1200
        nd = O.NodeTable(n)
        side = 50.0 * np.sqrt(np.pi * n / 20.0)
        nd.x, nd.y, nd.z = rng.uniform(0, side, n), rng.uniform(0, side, n), rng.uniform(0, 20, n)
        nd.rxprob[:] = np.where(rng.random(n) < 0.5, 1.0, rng.uniform(0, 1, n))
        nd.txprob[:] = np.where(rng.random(n) < 0.5, 1.0, rng.uniform(0, 1.1, n))
        ticks = []
        for k in range(3):
            src = np.sort(rng.choice(n, 50, replace=False))
            ticks.append((k * 1000, nd.packets(src, k * 1000, 8128)))
        return nd, "udgm", {"udgm_success_ratio_rx": 0.8}, None, 20260101, ticks
    if name == "const_lattice":
        g = np.arange(0, 30) * 10.0
        xx, yy = np.meshgrid(g, g)
        nd = O.NodeTable(xx.size)
        nd.x, nd.y = xx.ravel().copy(), yy.ravel().copy()
        src = np.arange(0, nd.n, 17)
        return nd, "udgm_const", {}, None, None, [(0, nd.packets(src, 0, 320))]
    if name == "udgm_lattice":
        # receivers AT the range: a 10 m lattice, range 50 -- (50, 0), (30, 40), (40, 30) and their mirror images are heard
        # (UDGMRadioMedium.java:76: ratio == 1 is in) exactly as long as Math.pow(d, 2.0) and Math.pow(50, 2.0) both come out
        # 2500 to the last bit: the scenario that pins the one JDK assumption (tests/test_oracle_pow_ulp.py) on a real JVM
        g = np.arange(0, 24) * 10.0
        xx, yy = np.meshgrid(g, g)
        nd = O.NodeTable(xx.size)
        nd.x, nd.y = xx.ravel().copy(), yy.ravel().copy()
        nd.rxprob[::5] = 0.75
        src = np.arange(5, nd.n, 23)
        return nd, "udgm", {"udgm_success_ratio_rx": 0.5}, None, 7, [(0, nd.packets(src, 0, 8128))]
    if name == "n2n":
        n = 200
        nd = O.NodeTable(n)
        nd.x, nd.y = rng.uniform(0, 100, n), rng.uniform(0, 100, n)
        m = np.where(rng.random((n, n)) < 0.1, rng.uniform(0, 1.2, (n, n)), 0.0)
        nd.int_id[7] = -1
        nd.rxprob[20:40] = 0.6
        src = np.sort(rng.choice(n, 40, replace=False))
        return nd, "n2n", {}, m, 5, [(0, nd.packets(src, 0, 8128))]
    if name == "null":
        n = 300
        nd = O.NodeTable(n)
        nd.channel[::3] = 11
        nd.enabled[::7] = 0
        return nd, "null", {}, None, None, [(0, nd.packets([0, 1, 2, 299], 0, 320))]
    if name == "logdist_shadow":
        n = 2500
        nd = O.NodeTable(n)
        side = 50.0 * np.sqrt(np.pi * n / 20.0)
        nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
        src = np.sort(rng.choice(n, 80, replace=False))
        return nd, "logdist", {"ld_sigma_db": 4.0, "ld_seed": 0xC0FFEE}, None, None, [(0, nd.packets(src, 0, 8128))]
    if name == "logdist_sinr_overlap":
        n = 1800
        nd = O.NodeTable(n)
        side = 50.0 * np.sqrt(np.pi * n / 20.0)
        nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
        nd.channel[:] = 11 + rng.integers(0, 4, n)
        ticks = []
        for k in range(6):
            src = np.sort(rng.choice(n, 60, replace=False))
            pk = nd.packets(src, 0, 0)
            pk["start_us"] = k * 1000 + rng.integers(0, 1000, len(pk))
            pk["air_us"] = rng.choice([320, 2048, 8128], len(pk))
            ticks.append((k * 1000, pk))
        return nd, "logdist", {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 99}, None, None, ticks
    raise KeyError(name)


SCENARIOS = ["udgm_default", "udgm_stochastic", "udgm_lattice", "const_lattice", "n2n", "null", "logdist_shadow",
             "logdist_sinr_overlap"]


def run_oracle(nd, kind, params, matrix, seed, ticks):
    kw = dict(params)
    if matrix is not None:
        kw["n2n_matrix"] = matrix
    mdl = O.model(KINDS[kind], **kw)
    state = O.lib().orc_jrandom_seed(seed) if seed is not None else 0
    sinr_mode = kind == "logdist" and params.get("ld_flags", 0) & 1
    onair = np.zeros(0, dtype=O.PACKET_DTYPE)
    outs = []
    for t0, new in ticks:
        onair = onair[onair["start_us"] + onair["air_us"] > t0] if sinr_mode else onair[:0]
        active = np.concatenate([onair, new])
        r = O.tick(mdl, nd, active, first_new=len(onair), rng_state=state)
        state = r.rng_state
        outs.append(r)
        onair = active
    return outs, state


def main():
    for name in SCENARIOS:
        nd, kind, params, matrix, seed, ticks = scenario(name)
        nd.as_struct()
        outs, state = run_oracle(nd, kind, params, matrix, seed, ticks)
        blob = {"kind": np.array(kind), "seed": np.array(-1 if seed is None else seed, dtype=np.int64),
                "n_ticks": np.array(len(ticks)), "final_rng_state": np.array(state, dtype=np.uint64),
                "param_names": np.array(sorted(params)), "param_values": np.array([params[k] for k in sorted(params)],
                                                                                   dtype=np.float64)}
        for f in NODE_FIELDS:
            blob["node_" + f] = getattr(nd, f)
        if matrix is not None:
            blob["matrix"] = matrix
        for i, ((t0, new), r) in enumerate(zip(ticks, outs)):
            blob["t%d_begin" % i] = np.array(t0, dtype=np.int64)
            blob["t%d_packets" % i] = new
            blob["t%d_pkt" % i] = r.pkt
            blob["t%d_dst" % i] = r.dst
            blob["t%d_verdict" % i] = r.verdict
            blob["t%d_rssi" % i] = r.rssi
            blob["t%d_sinr" % i] = r.sinr
            blob["t%d_interference" % i] = r.pkt_interference
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **blob)
        print("%-24s %6d nodes %2d ticks %7d heard links  %6.1f KB" % (
            name, nd.n, len(ticks), sum(r.count for r in outs), os.path.getsize(path) / 1024))

    # extension math: bit patterns
    rng = np.random.default_rng(2026)
    L = O.lib()
    xs = np.concatenate([np.exp(rng.uniform(-60, 60, 2000)), [1.0, 2.0, 10.0, 1e-10, 0.5, 1.4142135623730951]])
    ys = np.concatenate([rng.uniform(-40, 40, 2000), [0.0, 0.5, -0.5, 1.0, -10.0]])
    us = np.concatenate([rng.uniform(0, 1, 2000), [2.0 ** -53, 0.02425, 0.97575, 1 - 2.0 ** -53, 0.5]])
    lin = np.concatenate([10.0 ** rng.uniform(-28, 3, 500), [0.0, 2.0 ** -80, 2.0 ** -81, 1.0]])
    pairs = rng.integers(0, 2 ** 31 - 1, (500, 2)).astype(np.uint32)
    np.savez_compressed(
        os.path.join(HERE, "detmath.npz"),
        log2_x=xs, log2_y=np.array([L.orc_det_log2(v) for v in xs]),
        exp2_x=ys, exp2_y=np.array([L.orc_det_exp2(v) for v in ys]),
        normal_u=us, normal_g=np.array([L.orc_det_normal(v) for v in us]),
        fixed_x=lin, fixed_y=np.array([L.orc_fixed_roundtrip(v) for v in lin]),
        hash_pairs=pairs, hash_seed=np.array(12345, dtype=np.uint64),
        hash_h=np.array([L.orc_shadow_hash(12345, int(a), int(b)) for a, b in pairs], dtype=np.uint64))
    print("detmath ok")


if __name__ == "__main__":
    main()
