"""Time of the whole path to the hit table on the bench workload: filter cascade + domain definition (incl. clustered regions)
+ hit list, for the standard pipeline and for --fs.  Usage (GPU box): python tools/hits_bench.py [windows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bath_amd as ba
from bath_amd import synth

nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
path = os.path.join(ROOT, "tests", "golden", "Caudal_act.bhmm")
ctx = ba.Context(0)
hmm = ba.HMM(path)
om = ba.OProfile(ctx, ba.Profile(hmm))
flat, offsets, planted = synth.dna_windows(nwin, 1000, seed=42, hmm=hmm)
block = ba.SeqBlock(ctx, flat, offsets)
pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
import ctypes as C
for rep in range(3):                                      # the C ABI call alone, without the Python copies of the records
    st_, dmp, ndm, nsk = ba.PipelineStats(), C.POINTER(ba.FsDomain)(), C.c_int64(0), C.c_int64(0)
    t0 = time.perf_counter()
    rc = ba.lib().bath_hip_pipeline_hits(ctx._h, om._h, block._h, C.byref(pipe.params), 10.0, C.byref(st_), C.byref(dmp), C.byref(ndm), C.byref(nsk))
    t1 = time.perf_counter()
    print("bath_hip_pipeline_hits: %.1f ms (rc %d, %d domains)" % ((t1 - t0) * 1e3, rc, ndm.value), flush=True)
for rep in range(3):
    t0 = time.perf_counter(); stats, res = pipe.run(block); t1 = time.perf_counter()
    stats2, dm, nskip = pipe.run_hits(block); t2 = time.perf_counter()
    th = ba.TopHits(); th.add(dm, ["w%d" % i for i in range(nwin)], [1000] * nwin); th.finalize(stats2.nres, hmm.max_length); text = th.tblout(hmm.name, hmm.acc, hmm.M)
    t3 = time.perf_counter()
    print("std: cascade %.1f ms | cascade+domains %.1f ms (%d hits, %d clustered regions, %d past Fwd) | hit list+table %.1f ms (%d lines)" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, len(dm), nskip, stats2.n_past_fwd, (t3 - t2) * 1e3, text.count("\n")), flush=True)
om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct)); om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
pf = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
small = ba.SeqBlock(ctx, flat[: offsets[nwin // 4]], offsets[: nwin // 4 + 1])
for rep in range(2):
    t0 = time.perf_counter(); st, fw, dm, nskip = pf.run_frameshift_domains(om3, om5, small); t1 = time.perf_counter()
    print("--fs (%d windows): %.1f ms, %d DNA windows, %d hits, %d clustered regions" % (nwin // 4, (t1 - t0) * 1e3, len(fw), len(dm), nskip), flush=True)
