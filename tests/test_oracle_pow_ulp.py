"""CPU tier: a bound on the one JDK assumption the oracle takes on faith.

UDGMRadioMedium.java:69,74 square the distance and the range through Math.pow(v, 2.0).  The oracle (and the engine) restate
that as v * v: fdlibm's e_pow and HotSpot's intrinsic both special-case y == 2.  No JVM exists here to confirm it, and the
Java SE specification only promises a result within 1 ulp.  This test re-runs every golden scenario of the UDGM medium, the
K2 / K3 boundary lattice (and the golden scenario built on one, udgm_lattice) and the five BASELINE layouts through a test-only oracle entry (orc_udgm_pow_sensitivity) with
distanceSquared and distanceMaxSquared moved by one ulp either way -- everything a pow that is not special-cased may legally
return -- and holds the outcome to the committed table: how many heard / unheard verdicts flip, how many probabilities move
at all, and by how much at most.

What the table says: on the random layouts NOTHING flips (a flip needs d*d within one ulp of range*range: distances that are
doubles drawn from a continuum do not do that), and a reception probability moves by at most 4.5e-16 relative (two ulps) -- so far
below anything a java.util.Random draw can resolve (2^-53 steps) that a verdict could only change if a draw landed within
that sliver of p.  The ONLY verdicts that depend on the assumption are exact lattice hits, d == range to the last bit (K2,
the 3-4-5 lattice of the parity tests): with distanceSquared one ulp up or distanceMaxSquared one ulp down they turn from
heard (UDGMRadioMedium.java:76: ratio > 1 is out, ratio == 1 is in) to unheard.  integration/java/harness/
ReferenceParityHarness.java names these as the scenarios that would expose a JVM whose pow is not x * x."""
import numpy as np
import pytest

from golden_util import load
from util import KINDS

VARIANTS = ((1, 0), (-1, 0), (0, 1), (0, -1), (1, -1), (-1, 1))


def _golden(O, name):
    g = load(name)
    nd = O.NodeTable(g["nodes"].n)
    for f in ("x", "y", "z", "txpower", "channel", "enabled", "rxprob", "txprob", "int_id"):
        setattr(nd, f, getattr(g["nodes"], f).copy())
    pk = np.concatenate([t["packets"].astype(O.PACKET_DTYPE) for t in g["ticks"]])
    return nd, O.model(KINDS["udgm"], **g["params"]), pk


def _lattice(O, ratio_rx):
    """K2 / K3 of SURVEY.md 8c and the 3-4-5 lattice of tests/test_gpu_parity.py::test_boundary_lattice: receivers at integer
    offsets around a source, range 50 -- 20 of them at distance exactly 50 (50-0, 30-40, 14-48 and their mirror images)"""
    pts = [(dx, dy) for dx in range(-52, 53) for dy in range(-52, 53) if (dx, dy) != (0, 0)]
    nd = O.NodeTable(len(pts) + 1)
    nd.x[1:] = [p[0] for p in pts]
    nd.y[1:] = [p[1] for p in pts]
    return nd, O.model(KINDS["udgm"], udgm_success_ratio_rx=ratio_rx), nd.packets(np.array([0], dtype=np.int32), 0, 8128)


def _baseline(O, index, n, packets, ratio_rx):
    from radio_sim_amd import workload as W
    w = W.make_nodes(n, index, channels16=(index == 4))
    nd = O.NodeTable(n)
    nd.x, nd.y, nd.z, nd.channel = w.x, w.y, w.z, w.channel
    src = W.choose_sources(n, packets, 0xC0FFEE00 + index, 0)
    return nd, O.model(KINDS["udgm"], udgm_success_ratio_rx=ratio_rx), nd.packets(src, 0, 8128)


# (scenario, links evaluated, heard, and per variant (d2 ulps, dmax2 ulps): (flips, probabilities that move, largest relative move))
TABLE = {
    'golden udgm_default': (75136, 901, {(1, 0): (0, 0, 0.000e+00), (-1, 0): (0, 0, 0.000e+00), (0, 1): (0, 0, 0.000e+00), (0, -1): (0, 0, 0.000e+00), (1, -1): (0, 0, 0.000e+00), (-1, 1): (0, 0, 0.000e+00)}),
    'golden udgm_stochastic': (179850, 2758, {(1, 0): (0, 417, 2.462e-16), (-1, 0): (0, 377, 2.200e-16), (0, 1): (0, 423, 2.200e-16), (0, -1): (0, 461, 2.462e-16), (1, -1): (0, 844, 2.772e-16), (-1, 1): (0, 777, 3.334e-16)}),
    'golden udgm_lattice': (14375, 1654, {(1, 0): (226, 312, 1.850e-16), (-1, 0): (0, 370, 2.961e-16), (0, 1): (0, 682, 2.961e-16), (0, -1): (226, 656, 1.850e-16), (1, -1): (226, 868, 1.850e-16), (-1, 1): (0, 942, 4.441e-16)}),
    'lattice ratioRx=1.0': (11024, 7844, {(1, 0): (20, 0, 0.000e+00), (-1, 0): (0, 0, 0.000e+00), (0, 1): (0, 0, 0.000e+00), (0, -1): (20, 0, 0.000e+00), (1, -1): (20, 0, 0.000e+00), (-1, 1): (0, 0, 0.000e+00)}),
    'lattice ratioRx=0.5': (11024, 7844, {(1, 0): (20, 2480, 4.428e-16), (-1, 0): (0, 2684, 4.381e-16), (0, 1): (0, 3000, 4.381e-16), (0, -1): (20, 2688, 4.428e-16), (1, -1): (20, 4600, 4.428e-16), (-1, 1): (0, 4884, 4.441e-16)}),
    'BASELINE configs[0] layout (64 nodes, 1 packets) ratioRx=0.5': (63, 12, {(1, 0): (0, 5, 1.926e-16), (-1, 0): (0, 4, 1.793e-16), (0, 1): (0, 4, 1.793e-16), (0, -1): (0, 5, 3.333e-16), (1, -1): (0, 5, 3.852e-16), (-1, 1): (0, 8, 3.852e-16)}),
    'BASELINE configs[1] layout (10000 nodes, 100 packets) ratioRx=0.5': (999900, 2009, {(1, 0): (0, 662, 4.435e-16), (-1, 0): (0, 683, 4.407e-16), (0, 1): (0, 767, 4.407e-16), (0, -1): (0, 742, 4.435e-16), (1, -1): (0, 1199, 4.435e-16), (-1, 1): (0, 1188, 4.435e-16)}),
    'BASELINE configs[2] layout (100000 nodes, 64 packets) ratioRx=0.5': (6399936, 1176, {(1, 0): (0, 425, 4.434e-16), (-1, 0): (0, 410, 4.411e-16), (0, 1): (0, 450, 4.411e-16), (0, -1): (0, 455, 4.434e-16), (1, -1): (0, 750, 4.438e-16), (-1, 1): (0, 722, 4.438e-16)}),
    'BASELINE configs[3] layout (100000 nodes, 64 packets) ratioRx=0.5': (399814, 85, {(1, 0): (0, 27, 4.187e-16), (-1, 0): (0, 35, 4.214e-16), (0, 1): (0, 40, 4.214e-16), (0, -1): (0, 32, 4.187e-16), (1, -1): (0, 53, 4.303e-16), (-1, 1): (0, 64, 4.316e-16)}),
    'BASELINE configs[4] layout (1000000 nodes, 16 packets) ratioRx=0.5': (15999984, 326, {(1, 0): (0, 116, 4.403e-16), (-1, 0): (0, 123, 4.337e-16), (0, 1): (0, 131, 4.337e-16), (0, -1): (0, 128, 4.403e-16), (1, -1): (0, 211, 4.413e-16), (-1, 1): (0, 210, 4.405e-16)}),
}


def scenarios(O):
    yield "golden udgm_default", _golden(O, "udgm_default")
    yield "golden udgm_stochastic", _golden(O, "udgm_stochastic")
    yield "golden udgm_lattice", _golden(O, "udgm_lattice")
    yield "lattice ratioRx=1.0", _lattice(O, 1.0)
    yield "lattice ratioRx=0.5", _lattice(O, 0.5)
    for index, n, packets in ((1, 64, 1), (2, 10_000, 100), (3, 100_000, 64), (4, 100_000, 64), (5, 1_000_000, 16)):
        yield "BASELINE configs[%d] layout (%d nodes, %d packets) ratioRx=0.5" % (index - 1, n, packets), _baseline(O, index, n, packets, 0.5)


def measure(O):
    rows = {}
    for name, (nd, mdl, pk) in scenarios(O):
        per = {}
        ev = hd = None
        for v in VARIANTS:
            e, h, flips, moved, rel = O.udgm_pow_sensitivity(mdl, nd, pk, *v)
            assert ev in (None, e) and hd in (None, h)
            ev, hd = e, h
            per[v] = (flips, moved, rel)
        rows[name] = (ev, hd, per)
    return rows


def test_pow_within_one_ulp_changes_only_exact_lattice_hits(O):
    rows = measure(O)
    assert set(rows) == set(TABLE), "scenarios and table differ"
    for name, (ev, hd, per) in rows.items():
        want_ev, want_hd, want_per = TABLE[name]
        assert (ev, hd) == (want_ev, want_hd), name
        for v, (flips, moved, rel) in per.items():
            w = want_per[v]
            assert (flips, moved) == (w[0], w[1]), "%s %s" % (name, v)
            assert rel <= w[2] * 1.001 + 1e-300 and rel <= 4.5e-16, "%s %s: p moves by %.3g" % (name, v, rel)
        if "lattice" not in name:
            assert all(f == 0 for f, _, _ in per.values()), "%s: a verdict flipped on a layout without exact lattice hits" % name
    # the lattice: exactly the 20 points at d == 50 flip, and only when the ratio is pushed above 1
    for name in ("lattice ratioRx=1.0", "lattice ratioRx=0.5"):
        per = rows[name][2]
        assert [per[v][0] for v in VARIANTS] == [20, 0, 0, 20, 20, 0]


if __name__ == "__main__":   # prints the table (python tests/test_oracle_pow_ulp.py, from the repo root)
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import oracle as O_
    for name, (ev, hd, per) in measure(O_).items():
        print("    %r: (%d, %d, {%s})," % (name, ev, hd, ", ".join("%r: (%d, %d, %.3e)" % (v, *per[v]) for v in VARIANTS)))
