#!/usr/bin/env python3
"""Where a query's time goes in the configs[3] leg: every model's run_hits on the 12.5 Mb genome with the stage laps (BATH_HIP_TIMING=1)."""
import os, sys, time
os.environ["BATH_HIP_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bath_amd as ba
from bath_amd import synth, dist as bdist
ctx = ba.Context(0)
hmms, g, planted = bench.c4_genome(ba, synth, int(12.5e6))
for q, hmm in enumerate(hmms):
    om = ba.OProfile(ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
    wins = bdist.split_targets([len(g)], hmm.max_length)
    block = ba.SeqBlock(ctx, [g[s_:s_ + n] for _, s_, n, _ in wins]); block.set_context([c for _, _, _, c in wins])
    pipe.run_hits(block); pipe.run_hits(block)
    ctx.synchronize()
    sys.stderr.write("==== %s M=%d\n" % (hmm.name, hmm.M)); sys.stderr.flush()
    t0 = time.perf_counter()
    st, dm, nclust = pipe.run_hits(block)
    ms = (time.perf_counter() - t0) * 1e3
    pipe.run(block, want_results=False)
    stage = {n_: round(ms_, 3) for n_, ms_, _ in pipe.timings()}
    sys.stderr.write("     total %.2f ms, %d domains, %d past fwd, clustered %d; cascade kernels %s\n" % (ms, len(dm), st.n_past_fwd, nclust, stage)); sys.stderr.flush()
