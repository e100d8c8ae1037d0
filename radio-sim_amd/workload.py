"""Synthetic inputs of SURVEY.md section 8(d) / BASELINE.md (host logic, numpy only).

Positions: N points uniform in a square of side L = R*sqrt(pi*N/k), R = 50 m (the UDGM range,
UDGMRadioMedium.java:22), k = 20 expected in-range neighbours, z = 0, fp64, SplitMix64 with seed
0xC0FFEE00 + config index.  Node defaults as the reference (Transciever.java:11-18).
Concurrent Tx: per tick T = round(f*N) distinct sources by partial Fisher-Yates seeded with
(base seed, tick); payload 127 B => hex length 254 => air time 8128 us (RadioPacket.java:67-75);
tick = 1000 us (EmuLink.java:60).
"""
import math

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
TICK_US = 1000
AIR_US = 254 * 32

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def splitmix64(seed, count):
    """First `count` outputs of SplitMix64 seeded with `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * GOLDEN
        return _mix(state)


def to_unit(u64):
    """53-bit uniform in [0, 1)."""
    return (u64 >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


class NodeTable:
    """SoA node state with the reference's defaults."""

    def __init__(self, n):
        self.n = n
        self.x = np.zeros(n)
        self.y = np.zeros(n)
        self.z = np.zeros(n)
        self.txpower = np.zeros(n)                       # Transciever.java:11
        self.channel = np.full(n, 26, dtype=np.int32)    # :12
        self.enabled = np.ones(n, dtype=np.uint8)        # :13
        self.rxprob = np.ones(n)                         # :17
        self.txprob = np.ones(n)                         # :18
        self.int_id = np.arange(1, n + 1, dtype=np.int32)


CONFIGS = {
    # name: (config index, N, Tx fraction, model name)
    "c1": dict(index=1, n=64, tx=1, model="udgm_const"),
    "c2": dict(index=2, n=10_000, frac=0.01, model="logdist"),
    "c3": dict(index=3, n=100_000, frac=0.01, model="logdist_shadow"),
    "c4": dict(index=4, n=100_000, frac=0.05, model="logdist_sinr16"),
    "c5": dict(index=5, n=1_000_000, frac=0.001, model="logdist_sinr_overlap"),
}


def side_length(n, r=50.0, k=20.0):
    return r * math.sqrt(math.pi * n / k)


def make_nodes(n, config_index, r=50.0, k=20.0, channels16=False):
    seed = 0xC0FFEE00 + config_index
    u = to_unit(splitmix64(seed, 2 * n))
    side = side_length(n, r, k)
    nd = NodeTable(n)
    nd.x = u[0::2] * side
    nd.y = u[1::2] * side
    if channels16:
        # 802.15.4 channels 11..26: 11 + (splitmix(node) mod 16)
        h = _mix(np.arange(n, dtype=np.uint64) + np.uint64(seed))
        nd.channel = (11 + (h % np.uint64(16))).astype(np.int32)
    return nd


def choose_sources(n, t, base_seed, tick):
    """T distinct sources: partial Fisher-Yates over 0..n-1, SplitMix64 seeded (base_seed, tick)."""
    rnd = splitmix64((base_seed * 0x100000001B3 + tick) & 0xFFFFFFFFFFFFFFFF, t)
    swapped = {}
    out = np.empty(t, dtype=np.int32)
    for i in range(t):
        j = i + int(rnd[i] % np.uint64(n - i))
        vi = swapped.get(i, i)
        vj = swapped.get(j, j)
        out[i] = vj
        swapped[j] = vi
    return np.sort(out)


def tx_count(cfg):
    return cfg["tx"] if "tx" in cfg else int(round(cfg["frac"] * cfg["n"]))


def model_kwargs(name, seed=0xC0FFEE):
    """(kind name, params) of the build's bench configurations."""
    if name == "udgm_const":
        return "udgm_const", {}
    if name == "udgm":
        return "udgm", {}
    if name == "udgm_lossy":      # the reference's UDGM with successRatioRx < 1: every heard link draws
        return "udgm", dict(udgm_success_ratio_rx=0.9)
    if name == "logdist":
        return "logdist", {}
    if name == "logdist_shadow":
        return "logdist", dict(ld_sigma_db=4.0, ld_seed=seed)
    if name in ("logdist_sinr16", "logdist_sinr_overlap"):
        return "logdist", dict(ld_sigma_db=4.0, ld_seed=seed, flags=1)
    raise KeyError(name)
