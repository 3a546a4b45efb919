"""SSV kernel time by tile shape: BATH_HIP_SSV_WIDE=1 / BATH_HIP_SSV_NARROW=1 select the rules of bath_profile.hip.  Usage: python tools/ssv_shape_probe.py 409|459|1024"""
import sys, time, numpy as np, os
sys.path.insert(0, ".")
import bath_amd as ba
from bath_amd import synth
which = sys.argv[1]
if which == "1024":
    path = "/tmp/syn1024.bhmm"; synth.write_synthetic_bhmm(path, 1024, seed=1024); hmm = ba.HMM(path); nwin = 100000
elif which == "459":
    db = "tests/golden/tRNA-proteins.bhmm"
    hmm = [h for h in (ba.HMM(db, q) for q in range(ba.HMM.count(db))) if h.M == 459][0]; nwin = 200000
else:
    hmm = ba.HMM("tests/golden/MET-ct4.bhmm"); nwin = 200000
flat, _, _ = synth.dna_windows(nwin, 1000, seed=42, hmm=hmm)
off = np.arange(nwin + 1, dtype=np.int64) * 1000
ctx = ba.Context(0); om = ba.OProfile(ctx, ba.Profile(hmm)); dna = ba.SeqBlock(ctx, flat, off)
pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
st, _ = pipe.run(dna, want_results=False)
for _ in range(3): st, _ = pipe.run(dna, want_results=False)
t = {n: round(ms, 3) for n, ms, _ in pipe.timings()}
print("M=%d narrow=%s wide=%s ssv_f1 %.3f ms  n_past_msv %d n_past_fwd %d" % (hmm.M, os.environ.get("BATH_HIP_SSV_NARROW"), os.environ.get("BATH_HIP_SSV_WIDE"), t["ssv_f1"], st.n_past_msv, st.n_past_fwd))
