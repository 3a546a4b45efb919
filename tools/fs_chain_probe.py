#!/usr/bin/env python3
"""Probe of the strict 3-codon parsers: window-length distribution of the bench's --fs block, and the duration of a parser
launch on synthetic batches of equal-length windows (n windows of L nt), to separate a row pair's parallel part from its chain."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bath_amd as ba
from bath_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--dist", type=int, default=0)
ap.add_argument("--cases", default="32x1000,256x1000")
ap.add_argument("--bwd", type=int, default=0)
ap.add_argument("--model", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "Caudal_act.bhmm"))
args = ap.parse_args()
hmm = ba.HMM(args.model)
ctx = ba.Context(0)
om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
if args.dist:
    flat, offsets, planted = synth.dna_windows(1_000_000, 1000, seed=4242, hmm=hmm, frameshift=True)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    dna = ba.SeqBlock(ctx, flat, offsets)
    om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, dna, arrays=True)
    ln = np.sort(np.asarray(fw["length"]))[::-1]
    br = np.asarray(fw["length"])[np.asarray(fw["branch"]) == 1]
    print("windows", len(ln), "sum", int(ln.sum()), "max", ln[:8].tolist(), "p99 %d p90 %d p50 %d" % tuple(np.percentile(ln, [99, 90, 50])))
    if len(br): print("fs-branch windows", len(br), "max", np.sort(br)[::-1][:8].tolist(), "p50 %d" % np.percentile(br, 50))
    for b in (0, 32, 64, 128, 256, 512, 1024, 2048, 4096, 7000): print("rank", b, "length", int(ln[min(b, len(ln) - 1)]))
    bs = np.sort(br)[::-1]
    for b in (0, 16, 64, 128, 256, 512, 1024, 2000): print("fs-branch rank", b, "length", int(bs[min(b, len(bs) - 1)]))
rng = np.random.default_rng(7)
fn = ba.FS3BackwardParser if args.bwd else ba.FS3ForwardParser
for case in args.cases.split(","):
    n, L = (int(x) for x in case.split("x"))
    blk = ba.SeqBlock(ctx, [rng.integers(0, 4, size=L).astype(np.uint8) for _ in range(n)])
    fn(ctx, om3, blk, logsum=ba.LOGSUM_TABLE_SERIAL)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); fn(ctx, om3, blk, logsum=ba.LOGSUM_TABLE_SERIAL); best = min(best, time.perf_counter() - t0)
    print("M %d  %d windows x %d nt: %.3f ms  = %.2f us per row pair" % (hmm.M, n, L, best * 1e3, best * 1e6 / (L / 2)), flush=True)
