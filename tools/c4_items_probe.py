#!/usr/bin/env python3
"""configs[3]'s full job on one GPU, item by item: every (query, window group) item of bench.py's c4.full_job alone on the chip
(ms, with the stage laps of the slowest when BATH_HIP_TIMING=1), then the job's worker passes with each worker's wall time."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bath_amd as ba
from bath_amd import synth, dist as bdist
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
nwk = int(sys.argv[2]) if len(sys.argv) > 2 else 6
hmms = [ba.HMM(bench.DB, q) for q in range(ba.HMM.count(bench.DB))]
n_nt = int(mb * 1e6)
all_wins = [bdist.split_targets([n_nt], h.max_length) for h in hmms]
items = bdist.query_items_weighted([len(w) for w in all_wins], [sum(n for _, _, n, _ in w) * (h.M + 150.0) for w, h in zip(all_wins, hmms)], 1)
g, planted = synth.genome(n_nt, seed=4300, hmms=hmms, genes_per_model=max(4, n_nt // 400_000))
cost = lambda it: bdist.item_cost(hmms[it[0]].M, sum(n for _, _, n, _ in all_wins[it[0]][it[1]:it[2]]))
ctx = ba.Context(0)
def make(c, it):
    q, lo, hi = it
    om = ba.OProfile(c, ba.Profile(hmms[q]))
    pipe = ba.Pipeline(c, om, fs_pipe=False, ncbi_table=hmms[q].ct)
    blk = ba.SeqBlock(c, [g[s_:s_ + n] for _, s_, n, _ in all_wins[q][lo:hi]]); blk.set_context([cc for _, _, _, cc in all_wins[q][lo:hi]])
    pipe.run_hits(blk)
    return pipe, blk
for it in items:
    pipe, blk = make(ctx, it)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); st, dm, ncl = pipe.run_hits(blk, arrays=True); ts.append((time.perf_counter() - t0) * 1e3)
    print("item %-16s M %4d windows %4d..%4d alone %6.2f ms  domains %4d clustered %3d  cost %.2f" % (hmms[it[0]].name, hmms[it[0]].M, it[1], it[2], min(ts), len(dm), ncl, cost(it)), flush=True)
    del pipe, blk
own = bdist.deal([cost(it) for it in items], nwk)
wctx = [ba.Context(0) for _ in range(nwk)]
wjobs = [[make(wctx[w], it) + (it,) for it, o in zip(items, own) if o == w] for w in range(nwk)]
def wp(w, out):
    t0 = time.perf_counter()
    for pipe, blk, it in wjobs[w]:
        pipe.run_hits(blk, arrays=True)
    wctx[w].synchronize()
    out[w] = (time.perf_counter() - t0) * 1e3
for rep in range(4):
    out = [0] * nwk
    t0 = time.perf_counter()
    th = [threading.Thread(target=wp, args=(w, out)) for w in range(nwk)]
    [t.start() for t in th]; [t.join() for t in th]
    print("pass %d: %.2f ms; workers %s; items/worker %s" % (rep, (time.perf_counter() - t0) * 1e3, ["%.1f" % x for x in out], [[hmms[it[0]].name for _, _, it in j] for j in wjobs] if rep == 0 else ""), flush=True)
