"""Per-step wall time of the filter cascade on the bench block, for 1, 2 and 3 concurrent parts (BATH_HIP_LANES)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bath_amd as ba
from bath_amd import synth
path = os.path.join(ROOT, "tests", "golden", "Caudal_act.bhmm")
ctx = ba.Context(0)
hmm = ba.HMM(path)
om = ba.OProfile(ctx, ba.Profile(hmm))
flat, offsets, planted = synth.dna_windows(1000000, 1000, seed=42, hmm=hmm)
block = ba.SeqBlock(ctx, flat, offsets)
pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
for lanes in sys.argv[1:] or ["1", "2", "3"]:
    os.environ["BATH_HIP_LANES"] = lanes
    ts = []
    for i in range(14):
        t0 = time.perf_counter(); pipe.run(block, want_results=False) if "want_results" in pipe.run.__code__.co_varnames else pipe.run(block); ts.append((time.perf_counter() - t0) * 1e3)
    print("lanes", lanes, " ".join("%.1f" % t for t in ts), flush=True)
