#!/usr/bin/env python3
"""The --fs pass (configs[2]) with W worker contexts on one GPU, each owning a block and running whole passes on it at the same time
(the reference's worker threads, bathsearch.c:1119-1290).  Prints ms per block for each W and checks that every worker's domains are
those of the one-worker pass."""
import argparse, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bath_amd as ba
from bath_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--workers", type=str, default="1,2,3")
ap.add_argument("--passes", type=int, default=4, help="passes per worker")
ap.add_argument("--windows", type=int, default=1_000_000)
ap.add_argument("--distinct", action="store_true", help="every worker its own block (another seed) instead of copies of one block")
ap.add_argument("--stagger-ms", type=float, default=0.0, help="worker w starts its passes w x this many ms after worker 0 (are workers better off out of step?)")
ap.add_argument("--json", action="store_true", help="one JSON line at the end: {workers: {ms_per_block, domains_equal}} (bench.py reads it)")
args = ap.parse_args()
if os.environ.get("PROBE_TORCH") == "1":                        # what bench.py's process has done before its fs leg
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
if os.environ.get("PROBE_EXTRA_CTX") == "1":                    # ... and a context with a finished pass of its own, left alive
    _hmm = ba.HMM(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "Caudal_act.bhmm"))
    _f, _o = synth.dna_windows(200_000, 1000, seed=1, hmm=_hmm, frameshift=True)[:2]
    _c = ba.Context(0); _om = ba.OProfile(_c, ba.Profile(_hmm))
    _p = ba.Pipeline(_c, _om, fs_pipe=True, ncbi_table=_hmm.ct)
    _p.run_frameshift_domains(ba.FSOProfile(_c, ba.FSProfile(_hmm, 3, ncbi_table=_hmm.ct)), ba.FSOProfile(_c, ba.FSProfile(_hmm, 5, ncbi_table=_hmm.ct)), ba.SeqBlock(_c, _f, _o), arrays=True)
hmm = ba.HMM(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "Caudal_act.bhmm"))
KEYS = ("window", "strand", "ienv", "jenv", "iali", "jali", "ihmm", "jhmm", "n_shifted_codons")


def key(dm):
    return sorted(tuple(int(r[k]) for k in KEYS) + (int(np.float32(r["envsc"]).view(np.uint32)),) for r in dm)


blocks = {}
def block(seed):
    if seed not in blocks:
        blocks[seed] = synth.dna_windows(args.windows, 1000, seed=seed, hmm=hmm, frameshift=True)[:2]
    return blocks[seed]


ref = {}
summary = {}
for W in [int(x) for x in args.workers.split(",")]:
    objs = []
    for w in range(W):
        seed = 4242 + (w if args.distinct else 0)
        flat, offsets = block(seed)
        ctx = ba.Context(0)
        om = ba.OProfile(ctx, ba.Profile(hmm))
        om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
        om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
        dna = ba.SeqBlock(ctx, flat, offsets)
        pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
        pipe.run_frameshift_domains(om3, om5, dna, arrays=True)
        objs.append((ctx, om, om3, om5, dna, pipe, seed))
    for o in objs:
        o[0].synchronize()
    got = [None] * W

    def work(w):
        ctx, om, om3, om5, dna, pipe, seed = objs[w]
        if args.stagger_ms > 0 and w > 0:
            time.sleep(w * args.stagger_ms * 1e-3)
        for _ in range(args.passes):
            _, _, dm, _ = pipe.run_frameshift_domains(om3, om5, dna, arrays=True)
        ctx.synchronize()
        got[w] = dm

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(w,)) for w in range(W)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    got = [key(g) for g in got]
    same = []
    for w in range(W):
        seed = objs[w][6]
        if seed not in ref:
            ref[seed] = got[w]
        same.append(ref[seed] == got[w])
    print("workers %d: %d passes in %.1f ms = %.2f ms per block; domains %s; equal to first pass of that block: %s"
          % (W, W * args.passes, dt * 1e3, dt * 1e3 / (W * args.passes), [len(g) for g in got], same), flush=True)
    summary[str(W)] = {"ms_per_block": dt * 1e3 / (W * args.passes), "blocks": W * args.passes, "domains": [len(g) for g in got],
                       "domains_equal_to_first_pass_incl_envsc_bits": bool(all(same))}
    for o in objs:
        o[0].close()
if args.json:
    import json
    print(json.dumps(summary))
