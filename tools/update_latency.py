"""Cost of node changes between ticks (node-config-set: rm_node_update per node, rm_nodes_move for
the shim's dirty list) against a fresh snapshot upload, each followed by one tick of 1 % concurrent Tx.
Run on the GPU box:  python tools/update_latency.py [nodes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import radio_sim_amd as rsa
from radio_sim_amd import workload as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
nodes = W.make_nodes(n, 3)
kind, kw = rsa.MODEL_LOGDIST, W.model_kwargs("logdist_shadow")[1]
eng = rsa.Engine(0)
eng.upload_table(nodes)
eng.set_model(kind, **kw)
rng = np.random.default_rng(1)
srcs = np.sort(rng.choice(n, n // 100, replace=False)).astype(np.int32)
import ctypes as C
hip = C.CDLL("libamdhip64.so.7")      # the runtime the engine itself uses (no torch in this process)
d_src = C.c_void_p()
assert hip.hipMalloc(C.byref(d_src), C.c_size_t(srcs.nbytes)) == 0
assert hip.hipMemcpy(d_src, C.c_void_p(srcs.ctypes.data), C.c_size_t(srcs.nbytes), 1) == 0


def tick():
    eng.tick_run_sources_device(0, 1000, d_src.value, len(srcs), 0, W.AIR_US)
    return eng.result_count()[0]


def timed(label, prepare, reps=20):
    tick()
    t_upd = t_tick = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        prepare()
        t1 = time.perf_counter()
        tick()
        t2 = time.perf_counter()
        t_upd += t1 - t0
        t_tick += t2 - t1
    print("%-46s change %9.1f us   next tick %9.1f us   table builds so far %d"
          % (label, t_upd / reps * 1e6, t_tick / reps * 1e6, eng.receiver_table_builds()))


def one_update():
    i = int(rng.integers(0, n))
    nodes.x[i] += rng.normal(0, 2.0)
    eng.update_node(i, nodes.x[i], nodes.y[i], nodes.z[i], nodes.txpower[i], int(nodes.channel[i]), int(nodes.enabled[i]),
                    nodes.rxprob[i], nodes.txprob[i])


def walk(count):
    def f():
        who = rng.choice(n, count, replace=False).astype(np.int32)
        nodes.x[who] += rng.normal(0, 2.0, count)
        nodes.y[who] += rng.normal(0, 2.0, count)
        eng.move_nodes(who, nodes.x[who], nodes.y[who])
    return f


timed("no change", lambda: None)
timed("rm_node_update, one node", one_update)
timed("rm_nodes_move, 100 nodes (2 m random walk)", walk(100))
timed("rm_nodes_move, 1 %% of the nodes (%d)" % (n // 100), walk(n // 100))
timed("rm_nodes_move, 10 %% of the nodes (%d)" % (n // 10), walk(n // 10), reps=10)
timed("rm_nodes_upload (fresh snapshot, sorted again)", lambda: eng.upload_table(nodes), reps=5)
eng.close()
