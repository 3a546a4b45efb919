#!/usr/bin/env python3
"""Per-row-pair time of the strict 3-codon parsers on long windows of a long model (configs[4]: M = 1024, windows of ~9 kb):
kernel time / row pairs of the longest window, for n windows of length L."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bath_amd as ba
from bath_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=1024)
ap.add_argument("--L", type=int, default=8000)
ap.add_argument("--n", type=str, default="1,4,64,327")
args = ap.parse_args()
path = "/tmp/chain_probe_%d.bhmm" % args.M
synth.write_synthetic_bhmm(path, args.M, seed=args.M, name="p%d" % args.M)
hmm = ba.HMM(path)
ctx = ba.Context(0)
om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
rng = np.random.default_rng(1)
def times():
    arr = (ba.KernelTime * 32)()
    k = ba.lib().bath_hip_kernel_times(ctx._h, 32, arr)
    return {arr[i].name.decode(): float(arr[i].ms) for i in range(k)}
for n in [int(x) for x in args.n.split(",")]:
    wins = [rng.integers(0, 4, size=args.L).astype(np.uint8) for _ in range(n)]
    blk = ba.SeqBlock(ctx, wins)
    for backward, fn in ((False, ba.FS3ForwardParser), (True, ba.FS3BackwardParser)):
        fn(ctx, om3, blk, logsum=ba.LOGSUM_TABLE_SERIAL)
        t0 = time.perf_counter()
        fn(ctx, om3, blk, logsum=ba.LOGSUM_TABLE_SERIAL)
        dt = (time.perf_counter() - t0) * 1e3
        print("M %d L %d n %4d %s: %.1f ms wall = %.1f us per row pair = %.1f ns per node" % (args.M, args.L, n, "backward" if backward else "forward ", dt, dt * 1e3 / (args.L / 2), dt * 1e6 / (args.L / 2) / args.M), flush=True)
