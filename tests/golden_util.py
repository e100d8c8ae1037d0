"""Loading of the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENARIOS = ["udgm_default", "udgm_stochastic", "udgm_lattice", "const_lattice", "n2n", "null", "logdist_shadow",
             "logdist_sinr_overlap"]
NODE_FIELDS = ("x", "y", "z", "txpower", "channel", "enabled", "rxprob", "txprob", "int_id")


class Table:
    pass


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    nd = Table()
    for f in NODE_FIELDS:
        setattr(nd, f, z["node_" + f])
    nd.n = len(nd.x)
    params = {str(k): float(v) for k, v in zip(z["param_names"], z["param_values"])}
    for k in ("ld_flags", "ld_seed"):
        if k in params:
            params[k] = int(params[k])
    ticks = []
    for i in range(int(z["n_ticks"])):
        ticks.append(dict(begin=int(z["t%d_begin" % i]), packets=z["t%d_packets" % i], pkt=z["t%d_pkt" % i],
                          dst=z["t%d_dst" % i], verdict=z["t%d_verdict" % i], rssi=z["t%d_rssi" % i],
                          sinr=z["t%d_sinr" % i], interference=z["t%d_interference" % i]))
    seed = int(z["seed"])
    return dict(nodes=nd, kind=str(z["kind"]), params=params, matrix=z["matrix"] if "matrix" in z else None,
                seed=None if seed < 0 else seed, ticks=ticks, final_rng_state=int(z["final_rng_state"]))
