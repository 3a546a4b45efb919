#!/usr/bin/env python3
"""Per-pass wall times of the strict --fs pass on the bench block (one worker): min / median / max over N passes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bath_amd as ba
from bath_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
hmm = ba.HMM(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "Caudal_act.bhmm"))
flat, offsets = synth.dna_windows(1_000_000, 1000, seed=4242, hmm=hmm, frameshift=True)[:2]
ctx = ba.Context(0)
om = ba.OProfile(ctx, ba.Profile(hmm))
om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct)); om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
dna = ba.SeqBlock(ctx, flat, offsets)
pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
for _ in range(2): pipe.run_frameshift_domains(om3, om5, dna, arrays=True)
ts = []
for _ in range(n):
    t0 = time.perf_counter(); pipe.run_frameshift_domains(om3, om5, dna, arrays=True); ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
print("passes %d: min %.1f median %.1f mean %.1f max %.1f ms; sorted: %s" % (n, ts.min(), np.median(ts), ts.mean(), ts.max(), " ".join("%.0f" % t for t in np.sort(ts))))
