#!/usr/bin/env python3
"""bench.py -- the north-star measurement: the bathsearch filter cascade on MI355X.

Workload (BASELINE.json configs[1]): testsuite/Caudal_act.bhmm (M=145) against 10^6 synthetic 1 kb DNA
windows (iid ACGT, seed 42, 1% carrying a planted domain), both strands, six-frame translation ->
MSV/SSV -> bias -> Viterbi -> Forward, standard codon table, no --fs.  One "step" is one pass of the whole
cascade over the block, with the DNA already resident in HBM.

  value   = DNA residues searched per second, counted as the reference counts pli->nres
            (both strands: bathsearch.c:1073,1086), whole job over all ranks.
  roofline = the dominant kernel (ssv_orf_kernel, SSV over the length-sorted ORF list): algorithmic HBM bytes
            per launch / its device time, timed with HIP events on the library's own stream
            (bath_hip_pipeline_timings).
  cpu_baseline = the scalar C oracle (oracle/pipeline.c, a port of the same algorithms) on a bounded
            sample of the same windows, one process per host core.  A reported baseline, not the target.

N>1: launched by torch.distributed.run, one rank per GPU; the model is broadcast from rank 0 over RCCL,
every rank scores its own 10^6 windows (weak scaling), counters and hits are gathered on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MODEL = os.path.join(ROOT, "tests", "golden", "Caudal_act.bhmm")
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, MI355X_MICROARCH.md


_CPU_FLAT = None      # the DNA block, inherited by the forked baseline workers (never pickled)


def cpu_baseline_worker(args):
    """Scalar oracle cascade over a slice of windows (runs in a forked worker)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oracle_lib as ol
    length, lo, hi = args
    flat = _CPU_FLAT
    m = ol.Model(MODEL, 0)
    L = ol.lib()
    pli = ol.Pipeline()
    L.bo_pipeline_init(C.byref(pli), 0)
    res = C.POINTER(ol.OrfResult)()
    n, a = C.c_int(0), C.c_int(0)
    t0 = time.perf_counter()
    for w in range(lo, hi):
        d = ol.dsq_from(flat[w * length:(w + 1) * length])
        n.value = 0
        L.bo_pipeline_window(C.byref(pli), m.om, m.sd, C.byref(m.bg), ol.u8(m.basic), ol.u8(d), length,
                             C.byref(res), C.byref(n), C.byref(a))
    dt = time.perf_counter() - t0
    return dt, pli.nres, pli.cells_msv + pli.cells_vit + pli.cells_fwd, pli.pos_past_msv, pli.pos_past_fwd


def usable_cores(cap=32):
    """Cores this job may really use: affinity mask, cgroup quota, capped (a reported baseline, not a stress test)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline(flat, length, n_windows, budget_s=12.0):
    import multiprocessing as mp
    global _CPU_FLAT
    _CPU_FLAT = flat
    cores = usable_cores()
    # calibrate on one core (the worker times only its scoring loop), then size the sample for ~budget_s
    dt, *_ = cpu_baseline_worker((length, 0, 24))
    per_win = max(dt / 24, 1e-5)
    per_core = int(max(8, min(n_windows // cores, budget_s / per_win)))
    jobs = [(length, c * per_core, (c + 1) * per_core) for c in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        outs = pool.map(cpu_baseline_worker, jobs)
    wall = time.perf_counter() - t0
    busy = max(o[0] for o in outs)                     # slowest worker's scoring loop (excludes process start-up and model parsing)
    nres = sum(o[1] for o in outs)
    cells = sum(o[2] for o in outs)
    return {"value": nres / busy, "unit": "residues/s", "cores": cores, "kind": "port",
            "sample": "%d windows x %d nt (both strands) through the scalar C oracle (oracle/pipeline.c), %d processes, "
                      "%.1f s scoring (%.1f s wall incl. start-up)" % (per_core * cores, length, cores, busy, wall),
            "gcells_per_s": cells / busy / 1e9}, outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=1_000_000, help="DNA windows per GPU (BASELINE config: 10^6)")
    ap.add_argument("--length", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    import bath_amd as ba
    from bath_amd import dist as bdist, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the backend has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    # query model: read on rank 0, broadcast (RCCL over xGMI), parsed by every rank
    blob = open(MODEL, "rb").read() if rank == 0 else b""
    blob = bdist.broadcast_bytes(blob, 0, dev)
    tmp = "/tmp/bath_bench_model_%d.bhmm" % os.getpid()
    with open(tmp, "wb") as fh:
        fh.write(blob)
    hmm = ba.HMM(tmp)
    os.unlink(tmp)

    ctx = ba.Context(local_rank)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    flat, offsets, planted = synth.dna_windows(args.windows, args.length, seed=42 + rank, hmm=hmm)
    dna = ba.SeqBlock(ctx, flat, offsets)
    pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    stats = None
    for _ in range(args.warmup):
        stats, _ = pipe.run(dna, want_results=False)
    sync()
    t0 = time.perf_counter()
    stage_ms = {}
    stage_launches = {}
    for _ in range(args.steps):
        stats, _ = pipe.run(dna, want_results=False)
        for name, ms, nl in pipe.timings():
            stage_ms.setdefault(name, []).append(ms)
            stage_launches[name] = nl
    sync()
    elapsed = bdist.max_over_ranks(time.perf_counter() - t0, dev)

    # one more pass with the copy-out, to gather the hits on rank 0 (outside the timed region)
    stats, res = pipe.run(dna, want_results=True)
    lo = rank * args.windows
    merged = bdist.reduce_stats(stats, dev)
    hits = bdist.gather_results(res, lo, 0, dev)

    if rank == 0:
        tot = merged
        nres_step = tot["nres"]
        cells_step = tot["cells_msv"] + tot["cells_vit"] + tot["cells_fwd"]
        ms_step = elapsed / args.steps * 1e3
        value = nres_step / (elapsed / args.steps)
        # dominant kernel: ssv_orf_kernel.  A large block runs as <lanes> concurrent parts (bath_hip_pipeline_filters), so a
        # step launches the kernel <lanes> times, each on its part of the ORF list and overlapping the other parts' work.
        lanes = max(1, int(stage_launches.get("ssv_f1", 1)))
        k_ms = float(np.mean(stage_ms["ssv_f1"])) / lanes                 # average duration of one launch (HIP events, its own stream)
        orf_res = stats.cells_msv / hmm.M
        # 1 B per ORF residue + 16 B per ORF work-list record read once, ~34 B written per surviving ORF; per launch
        algo_bytes = (orf_res + 16.0 * stats.n_orfs + 34.0 * stats.n_past_msv) / lanes
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_ssv_orf_pmc.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch_full_block") / lanes
            except Exception:
                traffic = None
        out = {
            "metric": "DP Gcells/s + residues/s through bathsearch pipeline at 1/2/4/8 MI355X",
            "value": value, "unit": "residues/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "integer scores: u8 as exact binary16 (SSV), u8 (MSV), i16 (Viterbi); f32 (Forward, bias)", "data": "synthetic",
            "config": {"workload": "Caudal_act.bhmm (M=%d) vs %d x %d nt iid DNA windows per GPU (1%% planted), both strands, "
                                   "6-frame translation + MSV/bias/Viterbi/Forward filter cascade, codon table %d, no --fs"
                                   % (hmm.M, args.windows, args.length, hmm.ct),
                       "windows_per_gpu": args.windows, "window_nt": args.length, "M": hmm.M},
            "gcells_per_s": cells_step / (elapsed / args.steps) / 1e9,
            "cells_per_step": cells_step, "residues_per_step": nres_step,
            "stage_ms": {k: float(np.mean(v)) for k, v in stage_ms.items()},
            "survivors": {k: tot[k] for k in ("n_orfs", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd")},
            "hits_gathered": int(len(hits)) if hits is not None else 0,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "ssv_orf_kernel", "kernel_ms": k_ms,
                         "note": "DP rows held in VGPRs (integer scores as binary16): compulsory HBM traffic is 1 B per ORF residue, the kernel is bound by the issue rate of packed 16-bit VALU ops (see DESIGN.md 4.1); "
                                 "cell rate of this kernel = %.2f Tcells/s per launch while %d parts of the block overlap on separate streams"
                                 % (stats.cells_msv / lanes / (k_ms * 1e-3) / 1e12, lanes), "launches_per_step": lanes,
                         # the bound that does apply, for the reader: the issue rate of packed 16-bit VALU ops.  tools/valu_rate.hip
                         # measures 5.2e11 wave-instructions/s for v_pk_add_f16 / v_pk_maximum3_f16 on this chip at 8 waves per SIMD (4.4e11 at
                         # the 4 this kernel's registers allow; v_fma_f32: 9.3e11); a row costs 3 of them per 4 cells (2 adds + 1 maximum3),
                         # 64 lanes each: 5.2e11 x 64 x 4/3 = 44.4 Tcells/s for the chip; concurrent parts share it
                         "valu": {"tcells_per_s_per_launch": stats.cells_msv / lanes / (k_ms * 1e-3) / 1e12, "peak_tcells_per_s": 44.4,
                                  "frac_per_launch": stats.cells_msv / lanes / (k_ms * 1e-3) / 1e12 / 44.4, "concurrent_launches": lanes,
                                  "peak_source": "measured packed-op issue rate, tools/valu_rate.hip, profiles/r01_valu_rate.txt"}},
        }
        if not args.no_cpu_baseline and world == 1:           # the CPU baseline is measured on rank 0 of the 1-GPU run only
            base, _ = cpu_baseline(flat, args.length, args.windows)
            out["cpu_baseline"] = base
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
