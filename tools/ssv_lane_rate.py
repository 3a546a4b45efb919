"""Rate of the lane-per-target SSV kernel on an ORF-like amino-acid batch (measurement helper, not part of the product).

Geometric ORF lengths >= 20 (stop probability 3/64), sorted by length like the pipeline's work list.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bath_amd as ba

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9_000_000
rng = np.random.default_rng(1)
lens = 20 + rng.geometric(3 / 64, size=n) - 1
lens = np.minimum(lens, 333)
lens[::-1].sort()
off = np.zeros(n + 1, np.int64); np.cumsum(lens, out=off[1:])
flat = rng.integers(0, 20, size=int(off[-1]), dtype=np.uint8)
ctx = ba.Context(0)
hmm = ba.HMM(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "Caudal_act.bhmm"))
om = ba.OProfile(ctx, ba.Profile(hmm, 400))
sq = ba.SeqBlock(ctx, flat, off)
for it in range(3):
    t0 = time.time(); sc, st = ba.SSVFilter(ctx, om, sq); t1 = time.time()
    print("ssvfilter n=%d res=%d  %.2f ms  %.2f Tcells/s (M=%d)" % (n, off[-1], (t1 - t0) * 1e3, off[-1] * om.M / (t1 - t0) / 1e12, om.M))
