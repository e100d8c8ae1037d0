"""Summarise tools/as_rank_sweep.sh output: the whole tick on one GPU, every rank's share, the compute-side speed-up."""
import json
import sys
rows = [json.loads(ln) for ln in open(sys.argv[1]) if ln.startswith("{")]
whole, ranks = rows[0], rows[1:]
w = whole["ms_per_tick"] * 1e3
worst = max(r["ms_per_tick"] for r in ranks) * 1e3
print("whole tick, one GPU: %.3f us (%d ticks per launch, %d contexts)" % (w, whole["config"]["ticks_per_launch"], whole["config"]["contexts"]))
for i, r in enumerate(ranks):
    print("rank %d of %d: %.3f us per tick, %d heard links" % (i, len(ranks), r["ms_per_tick"] * 1e3, r["config"]["heard_links_last_tick"]))
print("slowest rank %.3f us -> compute-side strong scaling %.2fx on %d GPUs" % (worst, w / worst, len(ranks)))
