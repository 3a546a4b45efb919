#!/usr/bin/env python3
"""Cascade throughput with W worker contexts on one GPU (the reference's worker threads, bathsearch.c thread_loop: every worker
owns a block and its own pipeline object).  Each worker runs the cascade over its own resident block, --steps passes in total."""
import argparse, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bath_amd as ba
from bath_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--workers", type=int, default=2)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--windows", type=int, default=1_000_000)
args = ap.parse_args()
hmm = ba.HMM(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "Caudal_act.bhmm"))
flat, _, _ = synth.dna_windows(args.windows, 1000, seed=42, hmm=hmm)
offsets = np.arange(args.windows + 1, dtype=np.int64) * 1000
W = args.workers
objs = []
for w in range(W):
    ctx = ba.Context(0)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    dna = ba.SeqBlock(ctx, flat, offsets)
    pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
    pipe.run(dna, want_results=False); pipe.run(dna, want_results=False)
    objs.append((ctx, om, dna, pipe))
for o in objs: o[0].synchronize()
per = args.steps // W
out = [None] * W
def work(w):
    ctx, om, dna, pipe = objs[w]
    for _ in range(per):
        out[w], _ = pipe.run(dna, want_results=False)
    ctx.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(w,)) for w in range(W)]
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
print("workers %d: %d steps in %.2f ms = %.3f ms per step, %.3e residues/s; n_past_fwd %s" % (W, per * W, dt * 1e3, dt * 1e3 / (per * W), 2e9 * args.windows / 1e6 * per * W / dt / 1.0, [int(s.n_past_fwd) for s in out]))
