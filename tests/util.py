"""Shared helpers of the parity tests: run the same inputs through the CPU oracle and the HIP
engine (through its C ABI) and compare the heard-link lists."""
import os

import numpy as np


def sinr_lists_forced():
    """Do the run's developer knobs (tools/knob_sweep.sh) keep the SINR medium's lone ticks on the per-receiver lists instead of
    the scan (rm_airscan.hip)?"""
    return os.environ.get("RM_SINR_SCAN") == "0" or os.environ.get("RM_SINR_FRAMES") == "0" or os.environ.get("RM_FRAME_TICK") == "0"


KINDS = {"null": 0, "udgm": 1, "udgm_const": 2, "n2n": 3, "logdist": 4}

# oracle model field -> engine parameter field
_PARAM_MAP = {
    "udgm_success_ratio_tx": "udgm_success_ratio_tx", "udgm_success_ratio_rx": "udgm_success_ratio_rx",
    "udgm_transmission_range": "udgm_transmission_range", "udgm_interference_range": "udgm_interference_range",
    "const_range": "const_range", "ld_pl0_db": "ld_pl0_db", "ld_exponent": "ld_exponent", "ld_d0": "ld_d0",
    "ld_sigma_db": "ld_sigma_db", "ld_clip": "ld_clip", "ld_seed": "ld_seed",
    "ld_sensitivity_dbm": "ld_sensitivity_dbm", "ld_noise_dbm": "ld_noise_dbm", "ld_capture_db": "ld_capture_db",
    "ld_ifloor_dbm": "ld_ifloor_dbm", "ld_flags": "flags",
}


def to_tx_records(rsa, pkts):
    """oracle PACKET_DTYPE array -> rm_tx_record array."""
    pkts = np.atleast_1d(pkts)
    r = np.zeros(len(pkts), dtype=rsa.TX_RECORD_DTYPE)
    for f in ("x", "y", "z", "txpower", "txprob", "start_us", "air_us", "src", "channel"):
        r[f] = pkts[f]
    return r


def configure_engine(eng, nodes, kind, params, matrix=None):
    eng.upload_table(nodes)
    kw = {_PARAM_MAP[k]: v for k, v in params.items()}
    eng.set_model(KINDS[kind], **kw)
    if matrix is not None:
        eng.set_n2n_matrix(matrix)


def oracle_model(O, kind, params, matrix=None):
    kw = dict(params)
    if matrix is not None:
        kw["n2n_matrix"] = matrix
    return O.model(KINDS[kind], **kw)


def assert_same(res_gpu, res_cpu, what="", rtol=1e-5, exact_fp=True):
    assert res_gpu.count == res_cpu.count, "%s: heard links %d (gpu) vs %d (oracle)" % (what, res_gpu.count, res_cpu.count)
    np.testing.assert_array_equal(res_gpu.pkt, res_cpu.pkt, err_msg=what + " pkt")
    np.testing.assert_array_equal(res_gpu.dst, res_cpu.dst, err_msg=what + " dst")
    # the collision / delivery decision is bit-exact
    np.testing.assert_array_equal(res_gpu.verdict, res_cpu.verdict, err_msg=what + " verdict")
    # floating-point outputs: tolerance of the north star is 1e-5 relative; the design goal is bit-exact
    if exact_fp:
        np.testing.assert_array_equal(res_gpu.rssi, res_cpu.rssi, err_msg=what + " rssi")
        np.testing.assert_array_equal(res_gpu.sinr, res_cpu.sinr, err_msg=what + " sinr")
    else:
        np.testing.assert_allclose(res_gpu.rssi, res_cpu.rssi, rtol=rtol, atol=0, err_msg=what + " rssi")
        np.testing.assert_allclose(res_gpu.sinr, res_cpu.sinr, rtol=rtol, atol=1e-9, err_msg=what + " sinr")
    np.testing.assert_array_equal(res_gpu.pkt_interference, res_cpu.pkt_interference, err_msg=what + " tx failure")


def run_both(O, rsa, eng, nodes, kind, params, pkts, matrix=None, seed=None, first_new=0):
    """One tick of new frames `pkts` (no older on-air frames) through oracle and engine."""
    configure_engine(eng, nodes, kind, params, matrix)
    state = 0
    if seed is not None:
        eng.seed(seed)
        state = O.lib().orc_jrandom_seed(seed)
    mdl = oracle_model(O, kind, params, matrix)
    cpu = O.tick(mdl, nodes, pkts, first_new=first_new, rng_state=state)
    gpu = eng.tick(to_tx_records(rsa, pkts))
    return gpu, cpu


def random_nodes(O, n, side, seed, z_span=0.0):
    rng = np.random.default_rng(seed)
    nd = O.NodeTable(n)
    nd.x = rng.uniform(0, side, n)
    nd.y = rng.uniform(0, side, n)
    nd.z = rng.uniform(0, z_span, n) if z_span else np.zeros(n)
    return nd


class DeviceArray:
    """A device buffer through the HIP runtime the engine itself uses (ctypes; no torch in the process:
    torch wheels carry their own HIP/HSA runtime, and two runtimes in one process do not mix)."""

    def __init__(self, host_array=None, nbytes=None, device=None):
        import ctypes as C
        self._C = C
        self._hip = C.CDLL("libamdhip64.so.7")
        self.ptr = C.c_void_p()
        if device is not None:   # (multi-GPU tests: the buffer belongs to that device)
            assert self._hip.hipSetDevice(C.c_int(device)) == 0
        n = host_array.nbytes if host_array is not None else nbytes
        assert self._hip.hipMalloc(C.byref(self.ptr), C.c_size_t(max(n, 1))) == 0
        if host_array is not None:
            a = np.ascontiguousarray(host_array)
            assert self._hip.hipMemcpy(self.ptr, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0   # H2D

    @staticmethod
    def read(ptr, dtype, count):
        """device pointer -> numpy array (D2H through the same runtime)"""
        import ctypes as C
        hip = C.CDLL("libamdhip64.so.7")
        out = np.empty(count, dtype=dtype)
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(out.nbytes), 2) == 0   # D2H
        return out

    def free(self):
        if self.ptr:
            self._hip.hipFree(self.ptr)
            self.ptr = None
