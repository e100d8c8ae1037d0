"""One rank of a multi-GPU test (tests/test_gpu_multi.py): a FRESH process per GPU -- nothing here has touched a GPU before this
process picks its device -- that runs the library's own collective path (RCCL bound inside libradiomedium_hip.so; no torch
in the process) and leaves its share of the results in a directory for the parent to merge and compare with the oracle.

    python tests/multi/rank_worker.py <dir> <rank> <world>

The scenarios, in order (every rank runs the same sequence -- collectives must line up):
  batch      rm_dist_batch_run_sources_device: 6 ticks of the shadowed log-distance medium, one all-gather of source indices
  overlap    the same call with the SINR medium and frames that outlive their tick (two batches: the second begins with the
             first one's frames on the air) -- rm_airbatch.hip behind ONE all-gather per batch
  draws      rm_dist_tick_run_sources_device, UDGM with lossy links: per-packet draw counts (and the drawing links' nodes:
             regions interleave in node order) go round, every rank places its java.util.Random draws among the others'
  sinr_tick  rm_dist_tick_run_sources_device with the SINR medium, frames on the air across ticks
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def scenario_inputs(n, world):
    """what every rank and the parent agree on (seeded)"""
    rng = np.random.default_rng(20260104 + world)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    x, y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    rxprob = np.ones(n)
    rxprob[rng.choice(n, n // 4, replace=False)] = 0.7
    ticks = {name: [np.sort(rng.choice(n, t, replace=False)).astype(np.int32) for _ in range(k)]
             for name, (t, k) in {"batch": (120, 6), "overlap": (150, 10), "draws": (60, 3), "sinr_tick": (150, 4)}.items()}
    return x, y, rxprob, ticks


def pad(srcs, own, rank, slots):
    mine = srcs[own[srcs] == rank]
    out = np.full(slots, -1, dtype=np.int32)
    out[:len(mine)] = mine
    return out


def main():
    out_dir, rank, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import radio_sim_amd as rsa
    from radio_sim_amd import dist as D
    from util import DeviceArray
    n = 24_000
    x, y, rxprob, ticks = scenario_inputs(n, world)
    eng = rsa.Engine(rank)                       # one rank per device
    eng.upload_nodes(x, y)
    eng.set_partition_spatial(rank, world)
    own = eng.partition_of_nodes(world)
    # the communicator: rank 0 makes the id, the others find it in the directory
    id_file = os.path.join(out_dir, "unique_id.npy")
    if rank == 0:
        np.save(id_file + ".tmp.npy", rsa.Engine.comm_unique_id())
        os.replace(id_file + ".tmp.npy", id_file)
    for _ in range(600):
        if os.path.exists(id_file):
            break
        time.sleep(0.1)
    eng.comm_init_rank(np.load(id_file), world, rank)
    res = {}
    keep = []

    def save(name, b, r, n_pk):
        res["%s_%d_pkt" % (name, b)] = r.pkt
        res["%s_%d_dst" % (name, b)] = r.dst
        res["%s_%d_verdict" % (name, b)] = r.verdict
        res["%s_%d_rssi" % (name, b)] = r.rssi
        res["%s_%d_sinr" % (name, b)] = r.sinr
        res["%s_%d_pint" % (name, b)] = r.pkt_interference

    def slots_for(lists):
        return D.slots_needed(n, world, lists, own) + 2

    # ---- batch: the shadowed log-distance medium
    eng.set_model(rsa.MODEL_LOGDIST, ld_sigma_db=4.0, ld_seed=9)
    tk = ticks["batch"]
    slots = slots_for(tk)
    mine = np.stack([pad(s, own, rank, slots) for s in tk])
    d = DeviceArray(mine, device=rank)
    keep.append(d)
    t0 = np.arange(len(tk), dtype=np.int64) * 1000
    eng.dist_batch_run_sources_device(t0, t0 + 1000, d.ptr.value, slots, t0, 8128)
    for b in range(len(tk)):
        save("batch", b, eng.batch_result_copy(b, world * slots), world * slots)
    res["batch_slots"] = np.array([slots])

    # ---- overlap: SINR, frames of 8128 us over ticks of 1000 us, two batches of five
    eng.set_model(rsa.MODEL_LOGDIST, ld_sigma_db=4.0, ld_seed=9, flags=1)
    tk = ticks["overlap"]
    slots = slots_for(tk)
    res["overlap_slots"] = np.array([slots])
    for half in range(2):
        part = tk[half * 5:(half + 1) * 5]
        mine = np.stack([pad(s, own, rank, slots) for s in part])
        d = DeviceArray(mine, device=rank)
        keep.append(d)
        t0 = (100 + half * 5 + np.arange(5, dtype=np.int64)) * 1000
        eng.dist_batch_run_sources_device(t0, t0 + 1000, d.ptr.value, slots, t0, 8128)
        for b in range(5):
            save("overlap", half * 5 + b, eng.batch_result_copy(b, world * slots), world * slots)
    assert eng.air_batch_stats() == (2, 10)

    # ---- draws: UDGM, every heard link may consume a draw
    eng.upload_nodes(x, y, rxprob=rxprob)
    eng.set_partition_spatial(rank, world)
    eng.set_model(rsa.MODEL_UDGM, udgm_success_ratio_rx=0.9)
    eng.seed(77)
    tk = ticks["draws"]
    slots = slots_for(tk)
    res["draws_slots"] = np.array([slots])
    for b, s in enumerate(tk):
        d = DeviceArray(pad(s, own, rank, slots), device=rank)
        keep.append(d)
        eng.dist_tick_run_sources_device(300_000 + b * 1000, 301_000 + b * 1000, d.ptr.value, slots, 300_000 + b * 1000, 8128)
        save("draws", b, eng.result_copy(world * slots), world * slots)
        res["draws_%d_rng" % b] = np.array([eng.rng_state], dtype=np.uint64)

    # ---- sinr_tick: the SINR medium one tick at a time, frames on the air across ticks
    eng.upload_nodes(x, y)
    eng.set_partition_spatial(rank, world)
    eng.set_model(rsa.MODEL_LOGDIST, ld_sigma_db=4.0, ld_seed=9, flags=1)
    tk = ticks["sinr_tick"]
    slots = slots_for(tk)
    res["sinr_tick_slots"] = np.array([slots])
    for b, s in enumerate(tk):
        d = DeviceArray(pad(s, own, rank, slots), device=rank)
        keep.append(d)
        eng.dist_tick_run_sources_device(500_000 + b * 1000, 501_000 + b * 1000, d.ptr.value, slots, 500_000 + b * 1000, 8128)
        save("sinr_tick", b, eng.result_copy(world * slots), world * slots)

    res["own"] = own
    np.savez(os.path.join(out_dir, "rank%d.tmp.npz" % rank), **res)
    os.replace(os.path.join(out_dir, "rank%d.tmp.npz" % rank), os.path.join(out_dir, "rank%d.npz" % rank))
    for d in keep:
        d.free()
    eng.comm_destroy()
    eng.close()


if __name__ == "__main__":
    main()
