"""How much does a pipelined block loop gain?  Each lane runs its part of the bench block `reps` times back to back, so that one
part's tail overlaps the next block's translation + SSV.  Usage (GPU box): python tools/pipelined_probe.py"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bath_amd as ba
from bath_amd import synth

path = os.path.join(ROOT, "tests", "golden", "Caudal_act.bhmm")
ctx = ba.Context(0)
hmm = ba.HMM(path)
om = ba.OProfile(ctx, ba.Profile(hmm))
flat, offsets, _ = synth.dna_windows(1000000, 1000, seed=42, hmm=hmm)
dna = ba.SeqBlock(ctx, flat, offsets)
pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
L = ba.lib()
L.bath_hip_pipeline_filters_repeat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
for lanes in ("2", "3", "4"):
    os.environ["BATH_HIP_LANES"] = lanes
    for reps in (1, 10, 10):
        st = ba.PipelineStats()
        pipe.run(dna, want_results=False)
        ctx.synchronize()
        t0 = time.perf_counter()
        rc = L.bath_hip_pipeline_filters_repeat(ctx._h, om._h, dna._h, C.byref(pipe.params), reps, C.byref(st))
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print("lanes %s reps %2d: %.2f ms per block (rc %d, n_past_fwd %d)" % (lanes, reps, dt / reps * 1e3, rc, st.n_past_fwd), flush=True)
