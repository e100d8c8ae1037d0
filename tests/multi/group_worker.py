"""The one-process host of a multi-GPU test (tests/test_gpu_multi.py): rm_group_create with ONE MEMBER PER DEVICE -- a fresh
process, so that the devices are first touched here -- and the device-resident group tick over ncclCommInitAll
(rm_group_tick_run_sources_device: every member packs its transmitters' frames, one ncclGroupStart/End around the members'
ncclAllGather, every member sweeps its region).  The merged links are compared with the oracle in this process.

    python tests/multi/group_worker.py <devices>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    n_dev = int(sys.argv[1])
    import radio_sim_amd as rsa
    from oracle import oracle as O
    from util import DeviceArray
    n, t = 20_000, 150
    rng = np.random.default_rng(31)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd = O.NodeTable(n)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    nd.rxprob[rng.choice(n, n // 4, replace=False)] = 0.7
    grp = rsa.Group(list(range(n_dev)), spatial=True)
    try:
        grp.upload_table(nd)
        grp.set_model(rsa.MODEL_UDGM, udgm_success_ratio_rx=0.9)
        grp.seed(5)
        assert grp.uses_rccl(), "one member per device: the group's all-gather has to be RCCL's"
        from radio_sim_amd import dist as D
        own = D.owners(n, n_dev, positions=(nd.x, nd.y, nd.z))
        mdl = O.model(O.MODEL_UDGM, udgm_success_ratio_rx=0.9)
        state = O.lib().orc_jrandom_seed(5)
        keep = []
        for k in range(4):
            srcs = np.sort(rng.choice(n, t, replace=False)).astype(np.int32)
            slots = int(np.bincount(own[srcs], minlength=n_dev).max()) + 1
            ptrs = []
            order = []
            for r in range(n_dev):
                mine = srcs[own[srcs] == r]
                padded = np.full(slots, -1, dtype=np.int32)
                padded[:len(mine)] = mine
                d = DeviceArray(padded, device=r)   # (the members' source lists live on the members' own devices)
                keep.append(d)
                ptrs.append(d.ptr.value)
                order.append(padded)
            grp.tick_run_sources_device(k * 1000, k * 1000 + 1000, ptrs, slots, k * 1000, 8128)
            got = grp.result_copy(cap=1 << 21)
            order = np.concatenate(order)
            valid = np.nonzero(order >= 0)[0]
            cpu = O.tick(mdl, nd, nd.packets(order[valid], k * 1000, 8128), rng_state=state)
            state = cpu.rng_state
            assert got.count == cpu.count > 2000, (k, got.count, cpu.count)
            np.testing.assert_array_equal(got.pkt, valid[cpu.pkt])
            np.testing.assert_array_equal(got.dst, cpu.dst)
            np.testing.assert_array_equal(got.verdict, cpu.verdict)
            assert grp.rng_state == cpu.rng_state
        for d in keep:
            d.free()
        print("group of %d devices: 4 ticks with draws identical to the oracle" % n_dev)
    finally:
        grp.close()


if __name__ == "__main__":
    main()
