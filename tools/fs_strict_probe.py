#!/usr/bin/env python3
"""The bench's --fs block in strict mode: wall time per pass and device times per kernel (BATH_HIP_TIMING=1 adds the stages)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bath_amd as ba
from bath_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=1_000_000)
ap.add_argument("--strict", type=int, default=1)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--model", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "Caudal_act.bhmm"))
args = ap.parse_args()
hmm = ba.HMM(args.model)
flat, offsets, planted = synth.dna_windows(args.windows, 1000, seed=4242, hmm=hmm, frameshift=True)
ctx = ba.Context(0)
om = ba.OProfile(ctx, ba.Profile(hmm))
dna = ba.SeqBlock(ctx, flat, offsets)
om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
ctx.set_fs_strict(bool(args.strict))
pipe.run_frameshift_domains(om3, om5, dna, arrays=True)
kt = {}
t0 = time.perf_counter()
for _ in range(args.steps):
    stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, dna, arrays=True)
    for name, (ms, nl, cells, nbytes) in pipe.kernel_times().items():
        k = kt.setdefault(name, [0.0, 0.0])
        k[0] += ms / args.steps; k[1] += nl / args.steps
dt = (time.perf_counter() - t0) / args.steps
print(json.dumps({"strict": args.strict, "ms_per_pass": dt * 1e3, "domains": int(len(dm)), "fs_windows": int(len(fw)), "clustered_regions": int(nskip),
                  "kernels_ms": {k: round(v[0], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])}}))
