"""The --fs path to hits on a quarter of the bench block, 4 passes (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bath_amd as ba
from bath_amd import synth
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
path = os.path.join(ROOT, "tests", "golden", "Caudal_act.bhmm")
ctx = ba.Context(0); hmm = ba.HMM(path); om = ba.OProfile(ctx, ba.Profile(hmm))
flat, offsets, planted = synth.dna_windows(nwin, 1000, seed=42, hmm=hmm)
block = ba.SeqBlock(ctx, flat, offsets)
om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct)); om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
pf = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
for rep in range(4):
    t0 = time.perf_counter(); st, fw, dm, nskip = pf.run_frameshift_domains(om3, om5, block); t1 = time.perf_counter()
    print("--fs (%d windows): %.1f ms, %d DNA windows (%d frameshift branch), %d hits, %d clustered regions" % (nwin, (t1 - t0) * 1e3, len(fw), sum(1 for w in fw if w.branch == 1), len(dm), nskip), flush=True)

