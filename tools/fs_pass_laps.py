#!/usr/bin/env python3
"""The stage laps (BATH_HIP_TIMING=1) of N strict --fs passes with a PASS line after each: which stage is longer in the slow passes."""
import os, sys, time
os.environ["BATH_HIP_TIMING"] = "1"
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
for i in range(n + 2):
    t0 = time.perf_counter(); pipe.run_frameshift_domains(om3, om5, dna, arrays=True); ms = (time.perf_counter() - t0) * 1e3
    sys.stderr.write("PASS %d %.2f\n" % (i, ms)); sys.stderr.flush()
