"""One-off randomized sweep of the GPU path against the oracle (more seeds and models than the test suite runs).
Usage: python tools/stress_parity.py [n_seeds]   (GPU box; oracle must be built)."""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bath_amd as ba
import common
import oracle_lib as ol
import test_hits_gpu as TH
import test_fs_pipeline_gpu as TF

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = ba.Context(0)
models = [("PTH2.bhmm", 0), ("Caudal_act.bhmm", 0), ("AMP_N.bhmm", 0), ("MET-ct4.bhmm", 0), ("MET-ct4.bhmm", 1), ("2OG-FeII_Oxy_3.bhmm", 0)]
bad = 0
for name, idx in models:
    path = ol.GOLDEN + "/" + name
    model = ol.Model(path, idx)
    hmm = ba.HMM(path, idx)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    for seed in range(nseeds):
        rng = np.random.default_rng(1000 + 17 * seed + idx)
        wins = TF.frameshifted_windows(rng, model, n=30)
        try:
            stats, dm, nskip = TH.gpu_hits(ctx, path, idx, wins)
            pli, odm, per_d, onskip = model.run_pipeline_hits(wins)
            n1 = TH.compare_hits(dm, odm, per_d, nskip, onskip)
            pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
            st2, fw, dm2, nskip2 = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(ctx, wins))
            _, ofw, per_w, odm2, per_d2, oskip2 = model.run_pipeline_fsdom(wins)
            assert nskip2 == oskip2
            n2 = TF.compare_domains(model, dm2, odm2, per_d2, nskip2)
            print("ok   %-22s %d seed %d: %d std hits, %d fs-pipeline hits, skipped %d/%d" % (name, idx, seed, n1, n2, nskip, nskip2), flush=True)
        except Exception:
            bad += 1
            print("FAIL %-22s %d seed %d" % (name, idx, seed), flush=True)
            traceback.print_exc()
print("failures:", bad)
sys.exit(1 if bad else 0)
