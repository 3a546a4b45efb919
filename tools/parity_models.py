"""Cascade counters of the GPU path against the SSE2 CPU baseline over a large synthetic block, for every tutorial model (bench.py does
this for Caudal_act over 10^6 windows).  Usage (GPU box): python tools/parity_models.py [windows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench

nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
models = ["PTH2.bhmm", "AMP_N.bhmm", "MET-ct4.bhmm", "2OG-FeII_Oxy_3.bhmm", "Caudal_act.bhmm"]
cpu = {}
import bath_amd as ba
from bath_amd import synth
blocks = {}
for name in models:                                   # CPU legs first: they fork workers, before anything touches the GPU
    path = os.path.join(ROOT, "tests", "golden", name)
    bench.MODEL = path
    hmm = ba.HMM(path)
    flat, _, _ = synth.dna_windows(nwin, 1000, seed=7, hmm=hmm)
    base, counters, covered = bench.cpu_baseline(flat, 1000, nwin, budget_s=20.0)
    cpu[name] = (counters, covered)
    blocks[name] = flat
ctx = ba.Context(0)
bad = 0
for name in models:
    path = os.path.join(ROOT, "tests", "golden", name)
    hmm = ba.HMM(path)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    counters, covered = cpu[name]
    flat = blocks[name][:covered * 1000]
    offsets = np.arange(covered + 1, dtype=np.int64) * 1000
    for lanes in ("1", "2"):
        os.environ["BATH_HIP_LANES"] = lanes
        st, _ = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct).run(ba.SeqBlock(ctx, flat, offsets), want_results=False)
        diff = {f: (int(getattr(st, f)), counters[f]) for f in bench.COUNTERS if int(getattr(st, f)) != counters[f]}
        bad += bool(diff)
        print("%-22s M=%4d  %d windows, %s lane(s): %s  (past MSV %d, bias %d, Vit %d, Fwd %d)" % (name, hmm.M, covered, lanes, "EQUAL" if not diff else "DIFF %s" % diff,
              st.n_past_msv, st.n_past_bias, st.n_past_vit, st.n_past_fwd), flush=True)
print("mismatching runs:", bad)
