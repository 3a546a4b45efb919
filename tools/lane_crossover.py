#!/usr/bin/env python3
"""Cascade time by block size with the lane-per-ORF MSV / Viterbi kernels on (BATH_HIP_LANE_MIN_NT=0) and off (a huge value): where
the wave-per-ORF kernels stop being the better choice.  usage: lane_crossover.py <windows> [model index in tRNA-proteins, default Caudal]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bath_amd as ba
from bath_amd import synth
n = int(sys.argv[1])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hmm = ba.HMM(os.path.join(root, "tests", "golden", "Caudal_act.bhmm")) if len(sys.argv) < 3 else ba.HMM(os.path.join(root, "tests", "golden", "tRNA-proteins.bhmm"), int(sys.argv[2]))
flat, offsets = synth.dna_windows(n, 1000, seed=7, hmm=hmm)[:2]
ctx = ba.Context(0)
om = ba.OProfile(ctx, ba.Profile(hmm))
dna = ba.SeqBlock(ctx, flat, offsets)
pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
for _ in range(3): pipe.run(dna, want_results=False)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(10): st, _ = pipe.run(dna, want_results=False)
ctx.synchronize()
ms = (time.perf_counter() - t0) * 100
stage = {k: round(v, 3) for k, v, _ in pipe.timings()}
print("windows %d M %d LANE_MIN_NT %s: %.3f ms per pass; past_msv %d past_bias %d; msv %.3f vit %.3f" % (n, hmm.M, os.environ.get("BATH_HIP_LANE_MIN_NT", "default"), ms, st.n_past_msv, st.n_past_bias, stage["classify_msv"], stage["viterbi_windows"]))
