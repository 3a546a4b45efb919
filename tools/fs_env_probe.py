#!/usr/bin/env python3
"""Device times of the 5-codon envelope kernels on a batch of envelopes like the bench's --fs pass produces (4.8 k envelopes,
~200 nt on average, a few up to 750): run with BATH_HIP_FS_SERIAL=1 so that Forward and Backward do not share the chip."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bath_amd as ba
from bath_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4800)
ap.add_argument("--model", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "Caudal_act.bhmm"))
ap.add_argument("--strict", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
hmm = ba.HMM(args.model)
rng = np.random.default_rng(5)
basic = ba.gencode_basic(hmm.ct)
mat = synth.hmm_match_emissions(hmm)
envs = []
for e in range(args.n):
    nt = synth.frameshift_mutations(rng, synth.reverse_translate(rng, synth.sample_domain(rng, mat), basic))
    if e % 40 == 0:                                            # a few long ones: two domains and a spacer
        nt = np.concatenate([nt, rng.integers(0, 4, size=120, dtype=np.uint8), synth.frameshift_mutations(rng, synth.reverse_translate(rng, synth.sample_domain(rng, mat), basic))])
    envs.append(nt[:750])
ctx = ba.Context(0)
om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
blk = ba.SeqBlock(ctx, envs)
mode = ba.LOGSUM_TABLE_SERIAL if args.strict else ba.LOGSUM_TABLE


def times():
    arr = (ba.KernelTime * 32)()
    n = ba.lib().bath_hip_kernel_times(ctx._h, 32, arr)
    return {arr[i].name.decode(): float(arr[i].ms) for i in range(n)}


ba.FS5Envelopes(ctx, om5, blk, logsum=mode)
t0 = times()
for _ in range(args.reps):
    r = ba.FS5Envelopes(ctx, om5, blk, logsum=mode)
t1 = times()
L = np.array([len(e) for e in envs])
out = {"envelopes": args.n, "M": hmm.M, "rows": int(L.sum()), "max_rows": int(L.max()), "mean_rows": float(L.mean()), "cells": int((L + 1).sum() * (hmm.M + 1)),
       "strict": args.strict, "ms": {k: (t1[k] - t0.get(k, 0.0)) / args.reps for k in t1}, "fwd==bwd max abs": float(np.abs(r["fwdsc"] - r["bcksc"]).max())}
print(json.dumps(out))
