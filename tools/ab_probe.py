"""A/B timing of two builds of the library in alternating child processes on the same box.
Usage: python tools/ab_probe.py libA.so libB.so [rounds]   (each child: 1 and 2 concurrent parts, 12 steps, no result copy)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import bath_amd as ba
ba.LIB_PATH = sys.argv[1]
from bath_amd import synth
path = os.path.join(%r, "tests", "golden", "Caudal_act.bhmm")
ctx = ba.Context(0); hmm = ba.HMM(path); om = ba.OProfile(ctx, ba.Profile(hmm))
flat, offsets, planted = synth.dna_windows(1000000, 1000, seed=42, hmm=hmm)
block = ba.SeqBlock(ctx, flat, offsets)
pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
for lanes in (sys.argv[2],):
    os.environ["BATH_HIP_LANES"] = lanes.rstrip("c")
    os.environ["BATH_HIP_LANE_PRIO"] = "0" if lanes.endswith("c") else "1"      # "2c": two parts without stream priorities
    ts = []
    for i in range(13):
        t0 = time.perf_counter(); pipe.run(block, want_results=False); ts.append((time.perf_counter() - t0) * 1e3)
    ts = ts[1:]
    ssv = [ms / nl for name, ms, nl in pipe.timings() if name == "ssv_f1"]
    print("%%s lanes %%s: mean %%.2f min %%.2f max %%.2f  (ssv kernel %%.2f ms)" %% (os.path.basename(sys.argv[1]), lanes, np.mean(ts), min(ts), max(ts), ssv[0] if ssv else -1), flush=True)
os._exit(0)
''' % (ROOT, ROOT)
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rounds = [int(a) for a in sys.argv[1:] if a.isdigit()]
for r in range(rounds[0] if rounds else 3):
    for lanes in (os.environ.get("AB_LANES", "2,1").split(",")):
        for lib in libs:
            subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib), lanes], check=False)
