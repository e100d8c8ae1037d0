"""ctypes binding of libradiomedium_hip.so (include/radiomedium_hip.h).

The library is the product; nothing here computes anything.  Loading fails loudly if the
shared object is missing and cannot be built, and `rm_create` fails loudly without a gfx950
device -- there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_SO = os.environ.get("RM_LIBRARY") or os.path.join(_CSRC, "libradiomedium_hip.so")  # RM_LIBRARY: a diagnostic build (make stamps)

MODEL_NULL, MODEL_UDGM, MODEL_UDGM_CONST, MODEL_N2N, MODEL_LOGDIST = range(5)
UNHEARD, INTERFERED, DELIVERED = 0, 1, 2
LD_SINR = 1
MAX_BATCH = 512
RM_OK, RM_ERR_INVALID, RM_ERR_NO_DEVICE, RM_ERR_HIP, RM_ERR_CAPACITY, RM_ERR_STATE = 0, -1, -2, -3, -4, -5


class RadioMediumError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rm error %d: %s" % (code, message))
        self.code = code


class ModelParams(C.Structure):
    _fields_ = [("kind", C.c_int32), ("flags", C.c_int32),
                ("udgm_success_ratio_tx", C.c_double), ("udgm_success_ratio_rx", C.c_double),
                ("udgm_transmission_range", C.c_double), ("udgm_interference_range", C.c_double),
                ("const_range", C.c_double),
                ("ld_pl0_db", C.c_double), ("ld_exponent", C.c_double), ("ld_d0", C.c_double),
                ("ld_sigma_db", C.c_double), ("ld_clip", C.c_double), ("ld_seed", C.c_uint64),
                ("ld_sensitivity_dbm", C.c_double), ("ld_noise_dbm", C.c_double),
                ("ld_capture_db", C.c_double), ("ld_ifloor_dbm", C.c_double)]


class TxRecord(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double),
                ("txpower", C.c_double), ("txprob", C.c_double),
                ("start_us", C.c_int64), ("air_us", C.c_int64),
                ("src", C.c_int32), ("channel", C.c_int32)]


TX_RECORD_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("txpower", "<f8"), ("txprob", "<f8"),
                            ("start_us", "<i8"), ("air_us", "<i8"), ("src", "<i4"), ("channel", "<i4")])
assert TX_RECORD_DTYPE.itemsize == C.sizeof(TxRecord) == 64


class DeviceResult(C.Structure):
    _fields_ = [("count", C.c_void_p), ("pkt_offset", C.c_void_p), ("pkt", C.c_void_p), ("dst", C.c_void_p),
                ("verdict", C.c_void_p), ("rssi", C.c_void_p), ("sinr", C.c_void_p), ("capacity", C.c_uint32)]


class DenseResult(C.Structure):
    """rm_dense_result"""
    _fields_ = [("cell_mask", C.c_void_p), ("cell_count", C.c_void_p), ("count", C.c_void_p), ("pkt_offset", C.c_void_p),
                ("pkt_interference", C.c_void_p), ("n_packets", C.c_int32), ("chunks", C.c_int32), ("rx_first", C.c_int32)]


class KernelTime(C.Structure):
    """rm_kernel_time"""
    _fields_ = [("name", C.c_char * 96), ("stage", C.c_int32), ("launches", C.c_uint32), ("total_ms", C.c_double)]


class HostResult(C.Structure):
    _fields_ = [("count", C.c_uint32), ("n_packets", C.c_uint32), ("pkt_offset", C.c_void_p),
                ("pkt_interference", C.c_void_p), ("pkt", C.c_void_p), ("dst", C.c_void_p), ("verdict", C.c_void_p),
                ("rssi", C.c_void_p), ("sinr", C.c_void_p), ("pkt_rssi", C.c_void_p)]


class DeliveryView(C.Structure):
    _fields_ = [("count", C.c_uint32), ("pending_packets", C.c_uint32), ("oldest_packet", C.c_int64), ("packet", C.c_void_p), ("dst", C.c_void_p),
                ("rssi", C.c_void_p), ("n_runs", C.c_uint32), ("run_packet", C.c_void_p), ("run_first", C.c_void_p), ("run_count", C.c_void_p)]


class EvqOrder(C.Structure):
    _fields_ = [("top_start", C.c_int64), ("top_max", C.c_int64), ("ladders", C.c_int32), ("top_nonempty", C.c_int32)]


def library_path():
    return _SO


def build_library(force=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".cpp", ".h", ".hpp")) or f == "Makefile"]
    srcs.append(os.path.join(_CSRC, "..", "..", "include", "radiomedium_hip.h"))
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _CSRC, "-s", "-j4", "libradiomedium_hip.so"])
    return _SO


_lib = None

# name -> (restype, argtypes); every entry point declared in include/radiomedium_hip.h
SIGNATURES = {
    "rm_abi_version": (C.c_int, []),
    "rm_device_count": (C.c_int, []),
    "rm_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "rm_destroy": (None, [C.c_void_p]),
    "rm_last_error": (C.c_char_p, []),
    "rm_get_name": (C.c_char_p, [C.c_void_p]),
    "rm_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rm_model_defaults": (None, [C.POINTER(ModelParams), C.c_int32]),
    "rm_set_model": (C.c_int, [C.c_void_p, C.POINTER(ModelParams)]),
    "rm_get_model": (C.c_int, [C.c_void_p, C.POINTER(ModelParams)]),
    "rm_set_n2n_matrix": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "rm_set_base_rssi": (C.c_int, [C.c_void_p, C.c_double]),
    "rm_get_base_rssi": (C.c_double, [C.c_void_p, C.c_int32]),
    "rm_seed": (C.c_int, [C.c_void_p, C.c_int64]),
    "rm_get_rng_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rm_set_rng_state": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rm_nodes_upload": (C.c_int, [C.c_void_p, C.c_int32] + [C.c_void_p] * 9),
    "rm_node_update": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_int32, C.c_uint8, C.c_double, C.c_double]),
    "rm_receiver_table_builds": (C.c_int64, [C.c_void_p]),
    "rm_nodes_move": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rm_node_count": (C.c_int, [C.c_void_p]),
    "rm_set_partition": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rm_set_partition_spatial": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rm_partition_of_nodes": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "rm_region_split": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "rm_partition_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "rm_set_link_capacity": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rm_set_time": (C.c_int, [C.c_void_p, C.c_int64]),
    "rm_air_time_us": (C.c_int64, [C.c_int64]),
    "rm_event_times": (None, [C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rm_transmit": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_double),
                              C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                              C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]),
    "rm_tick_begin": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64]),
    "rm_enqueue_tx": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_double),
                                C.POINTER(C.c_int32)]),
    "rm_enqueue_tx_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "rm_tick_flush": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]),
    "rm_batch_result_view": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "rm_tick_flush_view": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rm_tick_run": (C.c_int, [C.c_void_p]),
    "rm_draws_pending": (C.c_int, [C.c_void_p]),
    "rm_draw_counts_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    "rm_draw_counts_to": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rm_tick_finish_draws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int]),
    "rm_draw_nodes_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "rm_tick_finish_draws_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int]),
    "rm_pack_tx_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    "rm_pack_tx_device_on": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    "rm_pack_tx_batch_device_on": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                             C.c_void_p]),
    "rm_tick_run_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32]),
    "rm_tick_run_records_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int64]),
    "rm_tick_run_sources_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int64,
                                             C.c_int64]),
    "rm_result_device": (C.c_int, [C.c_void_p, C.POINTER(DeviceResult)]),
    "rm_result_dense": (C.c_int, [C.c_void_p, C.POINTER(DenseResult)]),
    "rm_result_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rm_result_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                 C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]),
    "rm_sync": (C.c_int, [C.c_void_p]),
    "rm_batch_run_sources_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "rm_batch_run_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rm_batch_run_gathered_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "rm_batch_run_gathered_sources_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "rm_batch_run_gathered_blocks_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "rm_table_digest": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rm_batch_tile_reuse": (C.c_int, [C.c_void_p]),
    "rm_comm_available": (C.c_int, []),
    "rm_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "rm_comm_init_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "rm_comm_destroy": (C.c_int, [C.c_void_p]),
    "rm_comm_world": (C.c_int, [C.c_void_p]),
    "rm_comm_rank": (C.c_int, [C.c_void_p]),
    "rm_dist_batch_run_sources_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                                   C.c_int64]),
    "rm_dist_tick_run_sources_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_int64]),
    "rm_group_tick_run_sources_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_int64]),
    "rm_group_result_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]),
    "rm_group_uses_rccl": (C.c_int, [C.c_void_p]),
    "rm_batch_result_device": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(DeviceResult)]),
    "rm_batch_result_count": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rm_batch_result_copy": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]),
    "rm_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "rm_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]),
    "rm_profile_kernels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "rm_last_link_evaluations": (C.c_int64, [C.c_void_p]),
    "rm_slot_stats": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rm_air_ring_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rm_air_list_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rm_air_scan_ticks": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rm_air_batch_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rm_air_batch_pairs": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rm_group_create": (C.c_int, [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rm_group_destroy": (None, [C.c_void_p]),
    "rm_group_size": (C.c_int, [C.c_void_p]),
    "rm_group_set_partitioning": (C.c_int, [C.c_void_p, C.c_int32]),
    "rm_group_context": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "rm_group_set_model": (C.c_int, [C.c_void_p, C.POINTER(ModelParams)]),
    "rm_group_set_n2n_matrix": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "rm_group_seed": (C.c_int, [C.c_void_p, C.c_int64]),
    "rm_group_get_rng_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rm_group_set_link_capacity": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rm_group_nodes_upload": (C.c_int, [C.c_void_p, C.c_int32] + [C.c_void_p] * 9),
    "rm_group_node_update": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                       C.c_int32, C.c_uint8, C.c_double, C.c_double]),
    "rm_group_set_time": (C.c_int, [C.c_void_p, C.c_int64]),
    "rm_group_tick_begin": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64]),
    "rm_group_enqueue_tx": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_double),
                                      C.POINTER(C.c_int32)]),
    "rm_group_enqueue_tx_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "rm_group_tick_flush": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]),
    "rm_events_enable": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rm_events_disable": (C.c_int, [C.c_void_p]),
    "rm_events_next_packet": (C.c_int64, [C.c_void_p]),
    "rm_events_process": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "rm_node_info": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rm_node_info_changed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "rm_det_math": (C.c_double, [C.c_int32, C.c_double]),
    "rm_link_hash": (C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
    "rm_evq_init": (None, [C.c_void_p]),
    "rm_evq_add": (C.c_int32, [C.c_void_p, C.c_int64]),
    "rm_evq_drain": (None, [C.c_void_p, C.c_int64]),
    "rm_lcg_jump": (C.c_uint64, [C.c_uint64, C.c_uint64]),
    "rm_lcg_next_double": (C.c_double, [C.POINTER(C.c_uint64)]),
}


def lib():
    """Load (building first if needed) the HIP library.  Raises if it cannot be had."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build_library()
        L = C.CDLL(_SO)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError = the library does not match the header
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != RM_OK:
        raise RadioMediumError(code, lib().rm_last_error().decode("utf-8", "replace"))
    return code
