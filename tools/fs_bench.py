#!/usr/bin/env python3
"""BASELINE.json configs[2]: the same profile with --fs.  Frameshift stage + domain definition on filter survivors.

Synthetic block: <n> windows of 1 kb, every one carrying a gene emitted from the model with 0-3 single-nucleotide
insertions / deletions (both strands), so that the frameshift stage has a realistic amount of work per window; prints one
JSON line with the wall time per pass and the survivors of each step.  Not the headline benchmark (bench.py)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bath_amd as ba
from bath_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=20000)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--model", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "Caudal_act.bhmm"))
args = ap.parse_args()

hmm = ba.HMM(args.model)
rng = np.random.default_rng(7)
basic = ba.gencode_basic(hmm.ct)
mat = synth.hmm_match_emissions(hmm)
L = 1000
flat = rng.integers(0, 4, size=(args.windows, L), dtype=np.uint8)
for w in range(args.windows):
    nt = list(synth.reverse_translate(rng, synth.sample_domain(rng, mat), basic)[: L - 40])
    for _ in range(int(rng.integers(0, 4))):
        p = int(rng.integers(10, len(nt) - 10))
        if rng.random() < 0.5:
            del nt[p]
        else:
            nt.insert(p, int(rng.integers(0, 4)))
    nt = np.asarray(nt[: L - 2], dtype=np.uint8)
    pos = int(rng.integers(0, L - len(nt) + 1))
    if w % 2:
        nt = (3 - nt[::-1]).astype(np.uint8)
    flat[w, pos:pos + len(nt)] = nt
off = np.arange(args.windows + 1, dtype=np.int64) * L
ctx = ba.Context(0)
om = ba.OProfile(ctx, ba.Profile(hmm))
om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
dna = ba.SeqBlock(ctx, flat.reshape(-1), off)
pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
pipe.run_frameshift_domains(om3, om5, dna)
t0 = time.perf_counter()
for _ in range(args.steps):
    stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, dna)
dt = (time.perf_counter() - t0) / args.steps
fs_nt = sum(w.length for w in fw)
env_nt = sum(abs(d.jenv - d.ienv) + 1 for d in dm)
print(json.dumps({"workload": "%s (M=%d) --fs vs %d x %d nt windows with planted frameshifted genes" % (os.path.basename(args.model), hmm.M, args.windows, L),
                  "ms_per_pass": dt * 1e3, "residues_per_s": stats.nres / dt, "n_past_fwd_F4": stats.n_past_fwd,
                  "fs_windows": len(fw), "fs_window_nt": fs_nt, "fs_branch": sum(w.branch == 1 for w in fw), "std_branch": sum(w.branch == 2 for w in fw),
                  "domains": len(dm), "envelope_nt": env_nt, "multidomain_regions_skipped": nskip,
                  "fs3_parser_cells": fs_nt * hmm.M, "fs5_envelope_cells": env_nt * hmm.M,
                  "shifted_codons_found": int(sum(d.n_shifted_codons for d in dm))}))
