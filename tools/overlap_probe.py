"""Probe: do two pipeline calls on two HIP streams overlap usefully on one MI355X? (measurement helper)"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bath_amd as ba
from bath_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
hmm = ba.HMM(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "Caudal_act.bhmm"))
ctxs = [ba.Context(0), ba.Context(0)]
work = []
for k, ctx in enumerate(ctxs):
    om = ba.OProfile(ctx, ba.Profile(hmm))
    flat, off, _ = synth.dna_windows(n, 1000, 42 + k, hmm=hmm, ncbi_table=hmm.ct)
    dna = ba.SeqBlock(ctx, flat.reshape(-1), off)
    pipe = ba.Pipeline(ctx, om, ncbi_table=hmm.ct)
    pipe.run(dna, want_results=False)
    work.append((pipe, dna))

def run(k, reps):
    pipe, dna = work[k]
    for _ in range(reps):
        pipe.run(dna, want_results=False)

reps = 6
t0 = time.perf_counter(); run(0, reps); run(1, reps); t1 = time.perf_counter()
print("sequential: %.2f ms per call" % ((t1 - t0) / (2 * reps) * 1e3))
th = [threading.Thread(target=run, args=(k, reps)) for k in range(2)]
t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t1 = time.perf_counter()
print("two streams concurrently: %.2f ms per call" % ((t1 - t0) / (2 * reps) * 1e3))
