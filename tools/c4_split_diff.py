#!/usr/bin/env python3
"""Does a query's hit table depend on how its windows are cut into items?  One query of the configs[3] job (default: RtcB, 100 Mb)
as one block and as G groups of consecutive windows (in order and in reverse), finished like rank 0 does; prints the table lines
that differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import bath_amd as ba
from bath_amd import synth, dist as bdist
q = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mb = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
hmms = [ba.HMM(bench.DB, i) for i in range(ba.HMM.count(bench.DB))]
n_nt = int(mb * 1e6)
g, planted = synth.genome(n_nt, seed=4300, hmms=hmms, genes_per_model=max(4, n_nt // 400_000))
hmm = hmms[q]
wins = bdist.split_targets([n_nt], hmm.max_length)
ctx = ba.Context(0)
om = ba.OProfile(ctx, ba.Profile(hmm))
pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)


def run(lo, hi):
    blk = ba.SeqBlock(ctx, [g[s_:s_ + n] for _, s_, n, _ in wins[lo:hi]]); blk.set_context([c for _, _, _, c in wins[lo:hi]])
    st, dm, _ = pipe.run_hits(blk, arrays=True, nres_before=2 * sum(n - c for _, _, n, c in wins[:lo]))
    dm.rec["window"] += lo
    return st, dm


st, dm = run(0, len(wins))
whole = bench.finish_query_arrays(ba, hmm, dm, wins, st.nres, n_nt)
parts = [run(*bdist.shard_range(len(wins), k, G)) for k in range(G)]
nres = sum(int(p[0].nres) for p in parts)
fwd = bench.finish_query_arrays(ba, hmm, ba.HitArray.concat([p[1] for p in parts]), wins, nres, n_nt)
parts = [run(*bdist.shard_range(len(wins), k, G)) for k in range(G)]
rev = bench.finish_query_arrays(ba, hmm, ba.HitArray.concat([p[1] for p in parts[::-1]]), wins, nres, n_nt)
print("nres", int(st.nres), nres, "hits whole / groups in order / reversed:", whole[0], fwd[0], rev[0])
for name, t in (("in order", fwd), ("reversed", rev)):
    a, b = whole[1].splitlines(), t[1].splitlines()
    print(name, "identical text:", whole[1] == t[1], "same set of lines:", sorted(a) == sorted(b))
    for l in sorted(set(a) - set(b)): print("  only whole:", l[:230])
    for l in sorted(set(b) - set(a)): print("  only split:", l[:230])
# raw records near the first differing hit, and the pipeline counters of both runs
import re
only = sorted(set(fwd[1].splitlines()) ^ set(whole[1].splitlines()))
if only:
    nums = [int(x) for x in re.findall(r"\d{6,}", only[0])]
    pos = nums[1]
    st, dm = run(0, len(wins))
    parts = [run(*bdist.shard_range(len(wins), k, G)) for k in range(G)]
    print("counters whole:", {f: int(getattr(st, f)) for f in bdist.STAT_FIELDS})
    print("counters split:", {f: sum(int(getattr(p[0], f)) for p in parts) for f in bdist.STAT_FIELDS})
    for name, arrs in (("whole", [dm]), ("split", [p[1] for p in parts])):
        for a in arrs:
            r = a.rec
            off = np.array([w[1] for w in wins], dtype=np.int64)[r["window"]]
            near = np.abs(r["iali"] + off - pos) < 3000
            for x, o in zip(r[near], off[near]):
                print(name, "window", int(x["window"]), "iali", int(x["iali"] + o), "jali", int(x["jali"] + o), "ienv", int(x["ienv"] + o), "jenv", int(x["jenv"] + o), "bits", float(x["bitscore"]), "lnP", float(x["lnP"]), "reported", int(x["reported"]), "strand", int(x["strand"]))
