"""Probe: the filter cascade on a genome-shaped block (few windows of 256 kb) versus 1 kb windows of the same total size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bath_amd as ba

total = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
hmm = ba.HMM(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "Caudal_act.bhmm"))
ctx = ba.Context(0)
om = ba.OProfile(ctx, ba.Profile(hmm))
rng = np.random.default_rng(3)
flat = rng.integers(0, 4, size=total, dtype=np.uint8)
for wlen in (1000, 262144):
    n = total // wlen
    off = np.arange(n + 1, dtype=np.int64) * wlen
    dna = ba.SeqBlock(ctx, flat[: n * wlen], off)
    pipe = ba.Pipeline(ctx, om, ncbi_table=hmm.ct)
    pipe.run(dna, want_results=False)
    t0 = time.perf_counter()
    for _ in range(3):
        stats, _ = pipe.run(dna, want_results=False)
    dt = (time.perf_counter() - t0) / 3
    print("window %7d nt x %7d: %.2f ms/pass, %.3g residues/s, n_orfs %d past_msv %d past_fwd %d  stages %s" % (
        wlen, n, dt * 1e3, stats.nres / dt, stats.n_orfs, stats.n_past_msv, stats.n_past_fwd, {k: round(v, 2) for k, v, _ in pipe.timings()}))
    del dna
