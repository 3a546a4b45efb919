#!/usr/bin/env python3
"""bench.py -- the north-star measurement: the bathsearch filter cascade on MI355X.

Workload (BASELINE.json configs[1]): testsuite/Caudal_act.bhmm (M=145) against 10^6 synthetic 1 kb DNA
windows (iid ACGT, seed 42, 1% carrying a planted domain), both strands, six-frame translation ->
MSV/SSV -> bias -> Viterbi -> Forward, standard codon table, no --fs.  One "step" is one pass of the whole
cascade over the block, with the DNA already resident in HBM.

  value    = DNA residues searched per second, counted as the reference counts pli->nres
             (both strands: bathsearch.c:1073,1086), whole job over all ranks.
  roofline = the dominant kernel (ssv_orf_kernel, SSV over the length-sorted ORF list): algorithmic HBM bytes
             per launch / its device time, timed with HIP events on the library's own stream
             (bath_hip_pipeline_timings).  The kernel is bound by packed 16-bit VALU issue, not HBM:
             roofline.valu gives its cell rate from an un-overlapped (one part) pass against the measured issue ceiling.
  cpu_baseline = oracle/sse: an SSE2 128-bit striped restatement of the reference's impl_sse filters (same algorithms
             and layouts; the reference itself needs the un-vendored easel and cannot be built), bit-exact with the
             scalar oracle, driving the oracle's cascade: 1 thread and one process per host core.  It runs over the
             SAME windows as the GPU, so the run is also a full-block parity check (parity_full_block).
  fs       = BASELINE.json configs[2]: the same profile with --fs on a block whose planted domains carry indels and
             stops: whole path to hits; per-kernel device times and HBM fractions of the frameshift
             kernels (bath_hip_kernel_times).

N>1: `python bench.py --gpus N` starts N ranks itself (a fresh torch.distributed.run child, never a re-exec); under the
driver's own torch.distributed.run launch RANK is already set and this process is one rank.  The model is broadcast from
rank 0 over RCCL, every rank scores its own windows (weak: 10^6 per GPU; --scaling strong: 10^6 in total, sharded), counters
are reduced and the ORF records gathered on rank 0 (point to point, nothing replicated).
"""
import argparse
import json
import os
# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read once when the runtime starts.  The legs with
# many worker contexts launching small kernels at the same time (c4.concurrent_queries, c4.full_job: 9 host threads, a dozen streams)
# queue up behind one another on 4: 18.0 -> 14.5 ms per configs[3] database pass with 16; the cascade, the --fs pass and its two-worker
# leg are unchanged (10.7 / 67-68 / 57-59 ms with either).  INTEGRATION.md recommends the same to a host that runs several contexts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MODEL = os.path.join(ROOT, "tests", "golden", "Caudal_act.bhmm")
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, MI355X_MICROARCH.md
COUNTERS = ("nres", "n_orfs", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd", "pos_past_msv", "pos_past_bias", "pos_past_vit", "pos_past_fwd")

_CPU_FLAT = None      # the DNA block, inherited by the forked baseline workers (never pickled)


# ------------------------------------------------------------------------------------------------ CPU baseline (oracle/sse)

def cpu_baseline_worker(args):
    """The oracle's cascade on the SSE2 striped kernels over windows [lo, hi) (runs in a forked worker, no GPU state)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oracle_lib as ol
    length, lo, hi, sse = args
    flat = _CPU_FLAT
    m = ol.Model(MODEL, 0)
    L = ol.lib()
    L.bo_pipeline_use_sse(1 if sse else 0)
    pli = ol.Pipeline()
    L.bo_pipeline_init(C.byref(pli), 0)
    res = C.POINTER(ol.OrfResult)()
    n, a = C.c_int(0), C.c_int(0)
    d = np.full(length + 2, 255, dtype=np.uint8)
    dp = ol.u8(d)
    t0 = time.perf_counter()
    for w in range(lo, hi):
        d[1:length + 1] = flat[w * length:(w + 1) * length]
        n.value = 0
        L.bo_pipeline_window(C.byref(pli), m.om, m.sd, C.byref(m.bg), ol.u8(m.basic), dp, length, C.byref(res), C.byref(n), C.byref(a))
    dt = time.perf_counter() - t0
    return dt, {f: int(getattr(pli, f)) for f in COUNTERS}, int(pli.cells_msv + pli.cells_vit + pli.cells_fwd)


def usable_cores(cap=64):
    """Cores this job may really use: affinity mask, cgroup quota, capped (a reported baseline, not a stress test)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline(flat, length, n_windows, budget_s=30.0):
    """Runs BEFORE any GPU initialisation (forked workers must not inherit HIP state).  Returns the cpu_baseline object and
    the counters of the windows it covered (all of them when the cores allow it within the budget)."""
    import multiprocessing as mp
    global _CPU_FLAT
    _CPU_FLAT = flat
    cores = usable_cores()
    # one thread first: a prefix of the block, ~2 s
    n1 = min(n_windows, 4000)
    dt1, _, _ = cpu_baseline_worker((length, 0, n1, True))
    n1 = int(min(n_windows, max(n1, 2.0 / max(dt1 / n1, 1e-7))))
    dt1, c1, cells1 = cpu_baseline_worker((length, 0, n1, True))
    dts, cs, _ = cpu_baseline_worker((length, 0, min(n1, 2000), False))          # the scalar port, for the record
    per_win = dt1 / n1
    covered = int(min(n_windows, max(cores * 64, budget_s * cores / per_win * 0.8)))
    bounds = [covered * c // cores for c in range(cores + 1)]
    jobs = [(length, bounds[c], bounds[c + 1], True) for c in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        outs = pool.map(cpu_baseline_worker, jobs)
    wall = time.perf_counter() - t0
    busy = max(o[0] for o in outs)                     # slowest worker's scoring loop (excludes process start-up and model parsing)
    tot = {f: sum(o[1][f] for o in outs) for f in COUNTERS}
    cells = sum(o[2] for o in outs)
    base = {"value": tot["nres"] / busy, "unit": "residues/s", "cores": cores, "kind": "port",
            "label": "impl_sse-equivalent restatement, SSE2 128-bit (oracle/sse: striped SSV/MSV/Viterbi/Forward written from scratch, "
                     "bit-exact with the scalar oracle; the reference's own impl_sse needs the un-vendored easel and cannot be built here)",
            "sample": "%d of %d windows x %d nt (both strands), %d processes, %.1f s scoring (%.1f s wall incl. start-up)"
                      % (covered, n_windows, length, cores, busy, wall),
            "gcells_per_s": cells / busy / 1e9,
            "one_thread": {"value": c1["nres"] / dt1, "gcells_per_s": cells1 / dt1 / 1e9, "sample": "%d windows, %.1f s" % (n1, dt1)},
            "scalar_port_one_thread": {"value": cs["nres"] / dts, "sample": "%d windows through the scalar C oracle, %.1f s" % (min(n1, 2000), dts)}}
    return base, tot, covered


def fbits(x):
    return int(np.float32(x).view(np.uint32))


def fs_cpu_sample(flat, length, k):
    """Before any GPU initialisation: the oracle's --fs pipeline (generic_*_frameshift.c restated, scalar C) through domain
    definition over the FIRST <k> windows of the fs leg's block -- the DNA windows, their 3-codon Forward scores and every domain
    that the GPU's strict pass must reproduce on the same windows (fs.parity_check)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    model = ol.Model(MODEL, 0)
    t0 = time.perf_counter()
    pli, ofw, per_w, odm, per_d, oclust = model.run_pipeline_fsdom([flat[i * length:(i + 1) * length] for i in range(k)])
    wins = sorted((w, int(o.strand), int(o.n), int(o.length), fbits(o.fwdsc), int(o.branch)) for w, (a, b) in enumerate(per_w) for o in ofw[a:b])
    doms = sorted((w, int(o.ienv), int(o.jenv), int(o.iali), int(o.jali), int(o.ihmm), int(o.jhmm), int(o.n_shifted_codons), fbits(o.envsc), round(float(o.bitscore), 1))
                  for w, (a, b) in enumerate(per_d) for o in odm[a:b])
    return {"windows": k, "fs_windows": wins, "domains": doms, "clustered": int(oclust), "cpu_seconds": time.perf_counter() - t0,
            "counters": {f: int(getattr(pli, f)) for f in COUNTERS}}


_FS_CPU_JOB = None          # (model path, list of DNA windows, their contexts): inherited by the forked workers of fs_cpu_baseline


def fs_cpu_worker(span):
    """The oracle's whole --fs pipeline -- the cascade and both 3-codon parsers on the SSE2 striped kernels (oracle/sse: sse_filters.c,
    sse_fs.c), the envelope stage on the scalar restatement of generic_*_frameshift.c (5-codon Forward / Backward / decoding / optimal
    accuracy / null2, traceback, hits) -- over windows [lo, hi) of the job (a forked worker: no GPU state)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    path, seqs, ctxs = _FS_CPU_JOB
    lo, hi = span
    L_ = ol.lib()
    L_.bo_pipeline_use_sse(1)
    L_.bo_fs_use_sse(1)                                          # both 3-codon parsers: striped, probability space (oracle/sse/sse_fs.c)
    model = ol.Model(path, 0)
    model.fs(3); model.fs(5)                                     # profile construction is not part of the scoring loop
    t0 = time.perf_counter()
    pli, ofw, _, odm, _, _ = model.run_pipeline_fsdom(seqs[lo:hi], contexts=None if ctxs is None else ctxs[lo:hi])
    dt = time.perf_counter() - t0
    L_.bo_pipeline_use_sse(0)
    L_.bo_fs_use_sse(0)
    return dt, int(pli.nres), len(ofw), len(odm)


def fs_cpu_baseline(path, seqs, ctxs, what, probe=None, budget_s=12.0, min_per_s=0.0):
    """cpu_baseline of an --fs leg, BEFORE any GPU initialisation: every usable core scores its own slice of <seqs> through the
    oracle's --fs pipeline (fs_cpu_worker); the sample is sized from a one-thread probe so that the whole thing takes about
    <budget_s> seconds.  kind "port": the cascade and the 3-codon parsers are SSE2 striped restatements of impl_sse (fwdback_fs.c:97-533,
    :565-1050 in probability space); the 5-codon envelope stage is the SCALAR restatement of generic_*_frameshift.c (the reference runs
    that striped too: it would be faster than this figure on the envelopes)."""
    import multiprocessing as mp
    global _FS_CPU_JOB
    _FS_CPU_JOB = (path, seqs, ctxs)
    cores = usable_cores()
    n = len(seqs)
    probe = min(n, probe or max(1, n // 50))
    dt1, nres1, _, _ = fs_cpu_worker((0, probe))
    per = max(dt1 / probe, min_per_s)            # (min_per_s: a floor for jobs whose first units are not typical -- a genome window without a gene costs nothing)
    covered = int(min(n, max(cores, budget_s * cores / max(per, 1e-9) * 0.8)))
    cores = min(cores, covered)
    bounds = [covered * c // cores for c in range(cores + 1)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        outs = pool.map(fs_cpu_worker, [(bounds[c], bounds[c + 1]) for c in range(cores)])
    wall = time.perf_counter() - t0
    busy = max(o[0] for o in outs)
    _FS_CPU_JOB = None
    return {"value": sum(o[1] for o in outs) / busy, "unit": "residues/s", "cores": cores, "kind": "port",
            "label": "the oracle's --fs pipeline: SSE2 striped restatements of impl_sse's cascade and of the 3-codon frameshift parsers (probability space) + "
                     "SCALAR restatement of generic_*_frameshift.c for the 5-codon envelope stage (the reference runs that striped too, "
                     "impl_sse/fwdback_fs.c:2054-; it cannot be built here)",
            "sample": "%d of %d %s, %d processes, %.1f s scoring (%.1f s wall); %d DNA windows, %d domains"
                      % (covered, n, what, cores, busy, wall, sum(o[2] for o in outs), sum(o[3] for o in outs)),
            "one_thread": {"value": nres1 / dt1, "sample": "%d %s, %.1f s" % (probe, what, dt1)}}


def fs_parity_check(ba, ctx, pipe, om3, om5, flat, length, cpu):
    """The GPU's strict --fs pass over the windows the oracle scored: DNA windows with their Forward scores bitwise; the domains of
    the frameshift branch with exact coordinates and bitwise envelope scores; the standard branch's (fp32 odds-ratio arithmetic,
    its clustered regions sampled from matrices that agree to 1e-4) by coordinates with the unmatched ones counted."""
    k = cpu["windows"]
    blk = ba.SeqBlock(ctx, flat[:k * length], np.arange(k + 1, dtype=np.int64) * length)
    stats, fw, dm, nclust = pipe.run_frameshift_domains(om3, om5, blk, arrays=True)
    gw = sorted((int(r["window"]), int(r["strand"]), int(r["n"]), int(r["length"]), fbits(r["fwdsc"]), int(r["branch"])) for r in fw)
    branch_of = {(int(r["window"]), i): int(r["branch"]) for i, r in enumerate(fw)}
    g_all = [(int(r["window"]), int(r["ienv"]), int(r["jenv"]), int(r["iali"]), int(r["jali"]), int(r["ihmm"]), int(r["jhmm"]), int(r["n_shifted_codons"]),
              fbits(r["envsc"]), round(float(r["bitscore"]), 1), int(fw[int(r["fs_window"])]["branch"])) for r in dm]
    g_fs = sorted(t[:10] for t in g_all if t[10] == 1)
    g_std = sorted(t[:8] for t in g_all if t[10] != 1)
    c_all = list(cpu["domains"])
    c_left = list(c_all)
    fs_missing = 0
    for t in g_fs:                                                     # exact key + bitwise envsc (bitscore to the printed decimal rides along)
        m = [c for c in c_left if c[:9] == t[:9]]
        if m:
            c_left.remove(m[0])
        else:
            fs_missing += 1
    c_std = sorted(c[:8] for c in c_left)
    std_equal = len(set(g_std) & set(c_std))
    return {"what": "the first %d windows through the oracle's --fs pipeline (scalar restatement of generic_*_frameshift.c) before GPU init, and through the GPU's strict pass" % k,
            "windows": k, "fs_windows": len(gw), "fs_windows_identical_incl_forward_score_bits_and_branch": gw == cpu["fs_windows"],
            "fs_branch_domains": len(g_fs), "fs_branch_domains_exact_incl_envsc_bits": fs_missing == 0 and len(g_fs) + len(g_std) == len(c_all),
            "std_branch_domains": len(g_std), "std_branch_domains_with_identical_coordinates": std_equal,
            "clustered_regions": {"gpu": int(nclust), "cpu": cpu["clustered"]},
            "counters_equal": all(int(getattr(stats, f)) == cpu["counters"][f] for f in COUNTERS), "cpu_seconds": cpu["cpu_seconds"]}


# ------------------------------------------------------------------------------------------------ configs[3] / configs[4], one-GPU slices

DB = os.path.join(ROOT, "tests", "golden", "tRNA-proteins.bhmm")      # the reference's tutorial/tRNA-proteins.bhmm: 12 query models, M = 56..459
C5_M = 1024


def c5_model_path():
    from bath_amd import synth
    path = "/tmp/bath_bench_synth%d.bhmm" % C5_M
    if not os.path.exists(path):
        synth.write_synthetic_bhmm(path, C5_M, seed=C5_M, name="synth%d" % C5_M)
    return path


def c4_genome(ba, synth, n_nt):
    hmms = [ba.HMM(DB, q) for q in range(ba.HMM.count(DB))]
    g, planted = synth.genome(n_nt, seed=4300, hmms=hmms, genes_per_model=max(4, n_nt // 400_000))
    return hmms, g, planted


def c5_genome(ba, synth, n_nt):
    hmm = ba.HMM(c5_model_path())
    g, planted = synth.genome(n_nt, seed=4400, hmms=[hmm], genes_per_model=max(8, n_nt // 400_000), frameshift=True)
    return hmm, g, planted


def c5_sample_indices(wins, planted, k):
    """The first <k> genome windows that hold a planted gene entirely (the same choice on the CPU and the GPU side)."""
    idx = []
    for i, (_, s0, n, c) in enumerate(wins):
        if any(p >= s0 + c and p + ln <= s0 + n for _, p, ln in planted):
            idx.append(i)
            if len(idx) == k:
                break
    return idx


def hit_key(w, d):
    """A hit as the tables print it: window, envelope / alignment / model coordinates, the bit score to its printed decimal."""
    f = (lambda k: d[k]) if isinstance(d, np.void) else (lambda k: getattr(d, k))       # a numpy record (arrays=True) or a ctypes struct
    return (int(w),) + tuple(int(f(k)) for k in ("ienv", "jenv", "iali", "jali", "ihmm", "jhmm")) + (round(float(f("bitscore")), 1),)


def c45_cpu_samples(args):
    """Before any GPU initialisation: the oracle's pipeline on the SSE2 striped kernels over the FIRST windows of both genomes --
    the counters bench's GPU legs must reproduce on the same windows (parity_check of c4 and c5)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import bath_amd as ba
    import oracle_lib as ol
    from bath_amd import dist as bdist, synth
    L_ = ol.lib()
    L_.bo_pipeline_use_sse(1)
    out = {"c4": [], "c4_hits": [], "c5": None}
    hmms, g, _ = c4_genome(ba, synth, int(args.c4_mb * 1e6))
    t0 = time.perf_counter()
    for q, hmm in enumerate(hmms):
        wins = bdist.split_targets([len(g)], hmm.max_length)[:args.c45_sample_windows]
        model = ol.Model(DB, q)
        pli, odm, per_d, oskip = model.run_pipeline_hits([g[s_:s_ + n] for _, s_, n, _ in wins], contexts=[c for _, _, _, c in wins])
        out["c4"].append({f: int(getattr(pli, f)) for f in COUNTERS})
        out["c4_hits"].append(sorted(hit_key(w, o) for w, (a, b) in enumerate(per_d) for o in odm[a:b]))
    out["c4_seconds"] = time.perf_counter() - t0
    hmm5, g5, planted5 = c5_genome(ba, synth, int(args.c5_mb * 1e6))
    wins = bdist.split_targets([len(g5)], hmm5.max_length)
    wins = [wins[i] for i in c5_sample_indices(wins, planted5, args.c5_sample_windows)]
    t0 = time.perf_counter()
    model = ol.Model(c5_model_path(), 0)
    pli, ofw, per_w, odm, per_d, oclust = model.run_pipeline_fsdom([g5[s_:s_ + n] for _, s_, n, _ in wins], contexts=[c for _, _, _, c in wins])
    out["c5"] = {f: int(getattr(pli, f)) for f in COUNTERS}
    out["c5_windows"] = sorted((w, int(o.strand), int(o.n), int(o.length), fbits(o.fwdsc), int(o.branch)) for w, (a, b) in enumerate(per_w) for o in ofw[a:b])
    out["c5_hits"] = sorted(hit_key(w, o) + (fbits(o.envsc),) for w, (a, b) in enumerate(per_d) for o in odm[a:b])
    out["c5_seconds"] = time.perf_counter() - t0
    L_.bo_pipeline_use_sse(0)
    # cpu_baseline of the configs[4] leg: genome windows (262 kb + context each) through the oracle's --fs pipeline, one per core and pass
    all_wins = bdist.split_targets([len(g5)], hmm5.max_length)
    out["c5_baseline"] = fs_cpu_baseline(c5_model_path(), [g5[s_:s_ + n] for _, s_, n, _ in all_wins], [c for _, _, _, c in all_wins],
                                         "genome windows of %d nt + context (both strands)" % bdist.BLOCK_LENGTH, probe=1, budget_s=10.0, min_per_s=0.5)
    return out


def c4_leg(ba, synth, bdist, ctx, args, cpu):
    """BASELINE configs[3] at its one-GPU slice: the 12 query models of tRNA-proteins.bhmm, one after the other (the loop per
    query of bathsearch.c:737), against a genome cut into the reference's windows with context (dist.split_targets): the
    plain pipeline through domain definition to hits."""
    hmms, g, planted = c4_genome(ba, synth, int(args.c4_mb * 1e6))
    per_model, tot_ms, tot_res, tot_cells, tot_hits = [], 0.0, 0, 0, 0
    parity, hits_cmp = [], []
    for q, hmm in enumerate(hmms):
        om = ba.OProfile(ctx, ba.Profile(hmm))
        pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
        wins = bdist.split_targets([len(g)], hmm.max_length)
        seqs = [g[s_:s_ + n] for _, s_, n, _ in wins]
        ctxs = [c for _, _, _, c in wins]
        if cpu is not None:                                     # the windows the CPU leg scored, counter by counter
            k = min(args.c45_sample_windows, len(wins))
            sub = ba.SeqBlock(ctx, seqs[:k]); sub.set_context(ctxs[:k])
            st, sdm, _ = pipe.run_hits(sub)
            parity.append(all(int(getattr(st, f)) == cpu["c4"][q][f] for f in COUNTERS))
            got = sorted(hit_key(d.window, d) for d in sdm)
            hits_cmp.append({"gpu": len(got), "cpu": len(cpu["c4_hits"][q]), "identical": len(set(got) & set(cpu["c4_hits"][q]))})
        block = ba.SeqBlock(ctx, seqs); block.set_context(ctxs)
        pipe.run_hits(block)
        steps = 2
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            st, dm, nskip = pipe.run_hits(block)
        ms = (time.perf_counter() - t0) / steps * 1e3
        pipe.run(block, want_results=False)
        stage = {n_: round(ms_, 3) for n_, ms_, _ in pipe.timings()}
        found = sum(1 for qq, p_, ln in planted if qq == q and any(d.reported and min(d.iali, d.jali) + wins[d.window][1] < p_ + ln and
                                                                   max(d.iali, d.jali) + wins[d.window][1] > p_ for d in dm))
        cells = st.cells_msv + st.cells_vit + st.cells_fwd
        per_model.append({"name": hmm.name, "M": hmm.M, "ms": ms, "hits": int(sum(d.reported for d in dm)), "planted_found": found,
                          "planted": sum(1 for qq, _, _ in planted if qq == q), "clustered_regions": int(nskip), "cascade_stage_ms": stage})
        tot_ms += ms; tot_res += st.nres; tot_cells += cells; tot_hits += int(sum(d.reported for d in dm))
    # The same database pass with the queries spread over worker contexts (a query's search is a dozen small launches and
    # host steps: one query at a time leaves the chip idle most of the time; bathsearch's own loop is serial per query, its
    # worker threads split the target -- on a GPU the queries are the parallelism that is left for a genome this small).
    import threading
    nw = int(os.environ.get("BATH_BENCH_C4_WORKERS", "9"))
    # items as in the N-rank leg: (query, group of consecutive windows), a query cut in proportion to its share of the work (the 459-node
    # model in two), dealt longest-first onto the least loaded worker; an item carries the residue count the query's search has
    # reached at its first window (nres_before), so its hits are those of the query searched as one block
    all_wins = [bdist.split_targets([len(g)], h.max_length) for h in hmms]
    qcost = [sum(n for _, _, n, _ in w) * (h.M + 150.0) for w, h in zip(all_wins, hmms)]
    items = bdist.query_items_weighted([len(w) for w in all_wins], qcost, 1)
    owner = bdist.deal([bdist.item_cost(hmms[q].M, sum(n for _, _, n, _ in all_wins[q][lo:hi])) for q, lo, hi in items], nw)
    workers = []
    for w in range(nw):
        c = ba.Context(GPU)
        jobs = []
        for (q, lo, hi), o in zip(items, owner):
            if o != w:
                continue
            hmm = hmms[q]
            om = ba.OProfile(c, ba.Profile(hmm))
            pipe = ba.Pipeline(c, om, fs_pipe=False, ncbi_table=hmm.ct)
            wins = all_wins[q][lo:hi]
            blk = ba.SeqBlock(c, [g[s_:s_ + n] for _, s_, n, _ in wins]); blk.set_context([cc for _, _, _, cc in wins])
            before = 2 * sum(n - cc for _, _, n, cc in all_wins[q][:lo])
            pipe.run_hits(blk, nres_before=before)
            jobs.append((q, before, pipe, blk))
        workers.append((c, jobs))
    csteps = 3
    conc_by_worker = [None] * nw

    def work(w):
        c, jobs = workers[w]
        for _ in range(csteps):
            mine = {}
            for q, before, pipe, blk in jobs:
                _, dm, _ = pipe.run_hits(blk, arrays=True, nres_before=before)
                mine[q] = mine.get(q, 0) + int(dm.rec["reported"].sum())
        c.synchronize()
        conc_by_worker[w] = mine

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(w,)) for w in range(nw)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    conc_ms = (time.perf_counter() - t0) / csteps * 1e3
    conc_hits = [sum(m.get(q, 0) for m in conc_by_worker) for q in range(len(hmms))]
    for c, jobs in workers:
        jobs.clear()
        c.close()
    return {"workload": "tRNA-proteins.bhmm (12 query models, M = 56..459), each against a %.1f Mb synthetic genome (1/8 of configs[3]'s 100 Mb) cut into "
                        "%d-nt windows with 3*max_length context; cascade + domain definition + hits per model" % (args.c4_mb, bdist.BLOCK_LENGTH),
            "ms_per_database_pass": tot_ms, "residues_per_s": tot_res / (tot_ms * 1e-3), "gcells_per_s": tot_cells / (tot_ms * 1e-3) / 1e9,
            "hits": tot_hits, "models": per_model,
            "concurrent_queries": {"workers": nw, "ms_per_database_pass": conc_ms, "residues_per_s": tot_res / (conc_ms * 1e-3),
                                   "hits_equal_to_serial_loop": conc_hits == [m["hits"] for m in per_model],
                                   "items": len(items),
                                   "what": "the 12 queries as %d (query, window group) items dealt out to %d worker contexts (threads), longest first, one after the other within a worker" % (len(items), nw)},
            "parity_check": None if cpu is None else {"what": "the 10 pipeline counters of the first %d windows of every model against the SSE2 striped CPU pipeline" % args.c45_sample_windows,
                                                      "all_equal": all(parity), "per_model": parity, "cpu_seconds": cpu["c4_seconds"],
                                                      "hits": {"what": "the hits of those windows (window, envelope / alignment / model coordinates, bit score to its printed decimal): GPU vs CPU pipeline",
                                                               "gpu": sum(h["gpu"] for h in hits_cmp), "cpu": sum(h["cpu"] for h in hits_cmp),
                                                               "identical": sum(h["identical"] for h in hits_cmp), "per_model": hits_cmp}}}


def c5_leg(ba, synth, bdist, ctx, args, cpu):
    """BASELINE configs[4] at its one-GPU slice: a synthetic 1024-node model with --fs (16 nodes per lane in the frameshift
    kernels, SSV split over several lanes per ORF, the wavefront's ring in global memory) against 1/8 of the 1 Gb genome."""
    hmm, g, planted = c5_genome(ba, synth, int(args.c5_mb * 1e6))
    om = ba.OProfile(ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    wins = bdist.split_targets([len(g)], hmm.max_length)
    seqs = [g[s_:s_ + n] for _, s_, n, _ in wins]
    ctxs = [c for _, _, _, c in wins]
    pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    parity = None
    c5_hits = None
    if cpu is not None:
        sidx = c5_sample_indices(wins, planted, args.c5_sample_windows)
        k = len(sidx)
        sub = ba.SeqBlock(ctx, [seqs[i] for i in sidx]); sub.set_context([ctxs[i] for i in sidx])
        st, fw0, dm0, _ = pipe.run_frameshift_domains(om3, om5, sub, arrays=True)
        parity = {f: (int(getattr(st, f)), cpu["c5"][f]) for f in COUNTERS}
        gw = sorted((int(r["window"]), int(r["strand"]), int(r["n"]), int(r["length"]), fbits(r["fwdsc"]), int(r["branch"])) for r in fw0)
        gh = sorted(hit_key(int(r["window"]), r) + (fbits(r["envsc"]),) for r in dm0)
        c5_hits = {"what": "DNA windows (coordinates, 3-codon Forward score bitwise, branch) and hits (coordinates, bit score to its printed decimal, envelope score bitwise) of the "
                           "first %d genome windows that hold a planted gene: GPU strict pass vs the oracle's --fs pipeline" % k,
                   "fs_windows": {"gpu": len(gw), "cpu": len(cpu["c5_windows"]), "identical": gw == cpu["c5_windows"]},
                   "hits": {"gpu": len(gh), "cpu": len(cpu["c5_hits"]), "identical": len(set(gh) & set(cpu["c5_hits"]))}}
    block = ba.SeqBlock(ctx, seqs); block.set_context(ctxs)
    pipe.run_frameshift_domains(om3, om5, block, arrays=True)
    steps = 2
    kt = {}
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, block, arrays=True)
        for name, (ms, nl, cells, nbytes) in pipe.kernel_times().items():
            k = kt.setdefault(name, {"ms": 0.0, "launches": 0.0})
            k["ms"] += ms / steps; k["launches"] += nl / steps
    dt = (time.perf_counter() - t0) / steps
    summary = {"fs_windows": int(len(fw)), "fs_branch": int((fw["branch"] == 1).sum()), "domains": int(len(dm)), "reported": int(dm["reported"].sum()),
               "clustered_regions": int(nskip), "shifted_codons_found": int(dm["n_shifted_codons"].sum()), "planted": len(planted)}
    ctx.set_fs_strict(False)
    pipe.run_frameshift_domains(om3, om5, block, arrays=True)
    t0 = time.perf_counter()
    pipe.run_frameshift_domains(om3, om5, block, arrays=True)
    dtf = time.perf_counter() - t0
    ctx.set_fs_strict(True)
    # the cascade alone (F4 thresholds of the --fs pipeline), stage by stage
    casc = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    casc.run(block, want_results=False)
    t0 = time.perf_counter()
    for _ in range(3):
        cst, _ = casc.run(block, want_results=False)
    cms = (time.perf_counter() - t0) / 3 * 1e3
    stage = {n_: round(ms_, 3) for n_, ms_, _ in casc.timings()}
    ssv_ms = stage.get("ssv_f1", float("nan"))
    return {"workload": "synthetic %d-node model (bath_amd.synth.write_synthetic_bhmm, seed %d) --fs vs a %.0f Mb synthetic genome (1/8 of configs[4]'s 1 Gb), %d "
                        "windows of %d nt with context, %d planted frameshifted genes" % (C5_M, C5_M, args.c5_mb, len(wins), bdist.BLOCK_LENGTH, len(planted)),
            "ms_per_pass": dt * 1e3, "residues_per_s": stats.nres / dt, "mode": "strict (bit-identical frameshift recursions)",
            "fast": {"ms_per_pass": dtf * 1e3, "residues_per_s": stats.nres / dtf},
            "cpu_baseline": (cpu or {}).get("c5_baseline"),
            **summary, "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["ms"])},
            "cascade": {"ms": cms, "residues_per_s": cst.nres / (cms * 1e-3), "gcells_per_s": (cst.cells_msv + cst.cells_vit + cst.cells_fwd) / (cms * 1e-3) / 1e9,
                        "stage_ms": stage, "ssv_tcells_per_s": cst.cells_msv / (ssv_ms * 1e-3) / 1e12, "ssv_frac_of_packed_issue_peak": cst.cells_msv / (ssv_ms * 1e-3) / 1e12 / 44.4},
            "parity_check": None if parity is None else {"counters": "the 10 pipeline counters of the first %d windows that hold a planted gene (262144 nt each, --fs thresholds) against the CPU pipeline (SSE2 striped cascade + the oracle's frameshift stage)" % args.c5_sample_windows,
                                                         "all_equal": all(a == b for a, b in parity.values()),
                                                         "mismatches": {k: {"gpu": a, "cpu": b} for k, (a, b) in parity.items() if a != b}, "cpu_seconds": cpu["c5_seconds"], **(c5_hits or {})}}


def domain_records(domains):
    """A hit list as a sorted list of tuples -- every field that reaches the output, floats by their bits, the CIGAR string -- so that
    two searches can be compared record for record (fs_window, the index of the DNA window inside ONE pass, is local and left out)."""
    return sorted((int(d.window), int(d.strand), int(d.ienv), int(d.jenv), int(d.iali), int(d.jali), int(d.ihmm), int(d.jhmm), fbits(d.envsc), fbits(d.bitscore),
                   float(d.lnP), int(d.reported), int(d.n_shifted_codons), int(d.n_stops), d.cigar) for d in domains)


def records_diff(a, b, limit=3):
    """(equal, a few records only one side has) of two domain_records lists."""
    sa, sb = set(a), set(b)
    return a == b, {"n_ranks_search": len(a), "single_rank_search": len(b), "only_in_ranks_search": [list(map(str, x)) for x in sorted(sa - sb)[:limit]],
                    "only_in_single_rank_search": [list(map(str, x)) for x in sorted(sb - sa)[:limit]]}


def finish_query(ba, hmm, domains, wins, nres, genome_len):
    """Rank 0's end of a query (bathsearch.c:868-921): window coordinates -> target coordinates, E-values with the whole search's
    residue count, duplicates of the window overlaps removed, sorted, thresholded; returns (reported hits, --tblout text)."""
    for d in domains:
        off = wins[d.window][1]
        d.ienv += off; d.jenv += off; d.iali += off; d.jali += off
        d.window = 0
    th = ba.TopHits()
    th.add(domains, ["genome"], [genome_len])
    th.finalize(int(nres), hmm.max_length)
    return th.reported(), th.tblout(hmm.name, hmm.acc, hmm.M, show_cigar=True, show_header=False)      # (reported: duplicates of window overlaps not counted)


def finish_query_arrays(ba, hmm, hits, wins, nres, genome_len):
    """finish_query on a HitArray: the coordinate shift is four vector additions, the records go to the library as one array."""
    if hits is None or len(hits) == 0:
        hits = ba.HitArray(np.zeros(0, dtype=ba.FS_DOMAIN_DTYPE), b"")
    rec = hits.rec
    off = np.array([w[1] for w in wins], dtype=np.int64)[rec["window"]].astype(rec["ienv"].dtype) if len(rec) else 0
    for f in ("ienv", "jenv", "iali", "jali"):
        rec[f] += off
    rec["window"] = 0
    th = ba.TopHits()
    th.add_arrays(hits, ["genome"], [genome_len])
    th.finalize(int(nres), hmm.max_length)
    return th.reported(), th.tblout(hmm.name, hmm.acc, hmm.M, show_cigar=True, show_header=False)


def c4_leg_ranks(ba, synth, bdist, ctx, args, rank, world, dev, on_gpu, sync, total_mb):
    """BASELINE configs[3] over N ranks: the 12-model database is broadcast once; (query, window group) pairs are dealt to the
    ranks (dist.query_items_weighted / dist.deal); per query the hits travel to the query's owner rank (q mod N), which finishes it
    like the single-rank search (dist.exchange_query_hits, finish_query_arrays), the counters are all-reduced, and the finished
    tables are gathered on rank 0 (dist.gather_query_tables).  Strong scaling of ONE job: <total_mb> Mb x 12 queries."""
    import hashlib
    blob = open(DB, "rb").read() if rank == 0 else b""
    blob = bdist.broadcast_bytes(blob, 0, dev)                          # the whole database, once (1 MB over xGMI)
    tmp = "/tmp/bath_bench_db_%d.bhmm" % os.getpid()
    with open(tmp, "wb") as fh:
        fh.write(blob)
    hmms = [ba.HMM(tmp, q) for q in range(ba.HMM.count(tmp))]
    os.unlink(tmp)
    n_nt = int(total_mb * 1e6)
    all_wins = [bdist.split_targets([n_nt], h.max_length) for h in hmms]
    items = bdist.query_items_weighted([len(w) for w in all_wins], [sum(n for _, _, n, _ in w) * (h.M + 150.0) for w, h in zip(all_wins, hmms)], world)
    owner = bdist.deal([bdist.item_cost(hmms[q].M, sum(n for _, _, n, _ in all_wins[q][lo:hi])) for q, lo, hi in items], world)
    mine = [it for it, o in zip(items, owner) if o == rank]
    # A rank's items go to worker contexts of its GPU (host threads, one context each): a query's search is a chain of small launches
    # and host steps that leaves the chip mostly idle, so the items of a rank run side by side like the queries of c4.concurrent_queries
    # (BATH_BENCH_C4_RANK_WORKERS, default 6; 1: one after the other on the rank's main context).
    import threading
    cost = lambda it: bdist.item_cost(hmms[it[0]].M, sum(n for _, _, n, _ in all_wins[it[0]][it[1]:it[2]]))
    nwk = max(1, min(int(os.environ.get("BATH_BENCH_C4_RANK_WORKERS", "6")), len(mine)))
    wctx, wjobs = [], []
    if on_gpu:
        g, planted = synth.genome(n_nt, seed=4300, hmms=hmms, genes_per_model=max(4, n_nt // 400_000))
        wown = bdist.deal([cost(it) for it in mine], nwk)
        for w in range(nwk):
            c = ctx if nwk == 1 else ba.Context(GPU)
            jobs = []
            for (q, lo, hi), o in zip(mine, wown):
                if o != w:
                    continue
                om = ba.OProfile(c, ba.Profile(hmms[q]))
                pipe = ba.Pipeline(c, om, fs_pipe=False, ncbi_table=hmms[q].ct)
                blk = ba.SeqBlock(c, [g[s_:s_ + n] for _, s_, n, _ in all_wins[q][lo:hi]]); blk.set_context([cc for _, _, _, cc in all_wins[q][lo:hi]])
                before = 2 * sum(n - cc for _, _, n, cc in all_wins[q][:lo])    # what the query's search counted before this group (both strands)
                pipe.run_hits(blk, nres_before=before)
                jobs.append((q, lo, before, pipe, blk))
            wctx.append(c); wjobs.append(jobs)

    def worker_pass(w, out):
        res = []
        for q, lo, before, pipe, blk in wjobs[w]:
            st, dm, _ = pipe.run_hits(blk, arrays=True, nres_before=before)     # one record array + one CIGAR pool per item: no Python per hit
            dm.rec["window"] += lo
            res.append((q, dm, st))
        wctx[w].synchronize()
        out[w] = res

    def one_pass():
        outs = [None] * nwk
        if nwk == 1:
            worker_pass(0, outs)
        else:
            th = [threading.Thread(target=worker_pass, args=(w, outs)) for w in range(nwk)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        by_q, st_q = {}, {}
        for res in outs:
            for q, dm, st in res:
                by_q.setdefault(q, []).append(dm)
                acc = st_q.setdefault(q, dict.fromkeys(bdist.STAT_FIELDS, 0))
                for f in bdist.STAT_FIELDS:
                    acc[f] += int(getattr(st, f))
        return by_q, st_q

    from concurrent.futures import ThreadPoolExecutor
    finish_pool = ThreadPoolExecutor(max_workers=int(os.environ.get("BATH_BENCH_FINISH_THREADS", "4")))
    steps = 2
    sync()
    t0 = time.perf_counter()
    busy = t_gather = t_finish = 0.0
    for _ in range(steps):
        tb = time.perf_counter()
        if on_gpu:
            by_q, st_q = one_pass()
        else:                                                           # --plumbing-only: fabricated hits, the deal and the collectives are what runs
            by_q, st_q = {}, {}
            for q, lo, hi in mine:
                d = ba.FsDomain(); d.window = lo; d.reported = 1; d.iali = 10 + q; d.jali = 100 + q; d.ienv = 10 + q; d.jenv = 100 + q
                d.lnP = -50.0 - q; d.bitscore = 60.0 + q; d.cigar = "%dM" % (30 + q)
                by_q.setdefault(q, []).append(ba.HitArray.from_domains([d]))
                acc = st_q.setdefault(q, dict.fromkeys(bdist.STAT_FIELDS, 0))
                acc["nres"] += 2 * sum(n - c for _, _, n, c in all_wins[q][lo:hi])
        busy += time.perf_counter() - tb
        tg = time.perf_counter()
        # p7_tophits_Merge per query on the query's OWNER (q mod N), p7_pipeline_Merge for all queries in one all-reduce
        owned = bdist.exchange_query_hits({q: ba.HitArray.concat(v) for q, v in by_q.items()}, dev)
        merged = bdist.reduce_query_stats(st_q, len(hmms), dev)
        t_gather += time.perf_counter() - tg
        tf = time.perf_counter()
        # every query finished on its owner, inside the timed region (bathsearch.c:868-921: E-values with the whole search's residue
        # count, duplicates of the window overlaps, sort, thresholds, the table's text); a rank's queries are independent and the
        # work is the library's: a few threads, as bathsearch's output stage could.  Only the finished tables travel to rank 0.
        my_q = [q for q in range(len(hmms)) if bdist.query_owner(q, world) == rank]
        done = dict(zip(my_q, finish_pool.map(lambda q: finish_query_arrays(ba, hmms[q], owned.get(q), all_wins[q], merged[q]["nres"], n_nt), my_q)))
        t_finish += time.perf_counter() - tf
        tg = time.perf_counter()
        got_tables = bdist.gather_query_tables(done, 0, dev)
        t_gather += time.perf_counter() - tg
        tables = [got_tables[q] for q in range(len(hmms))] if rank == 0 else []
    sync()
    dt = bdist.max_over_ranks((time.perf_counter() - t0) / steps, dev)
    busy_all = bdist.gather_floats(busy / steps * 1e3, 0, dev)
    if on_gpu and nwk > 1:
        for jobs in wjobs:
            jobs.clear()
        for c in wctx:
            c.close()
    if rank != 0:
        return None
    out = {"workload": "tRNA-proteins.bhmm (12 query models, broadcast once) vs ONE %.0f Mb synthetic genome: (query, window group) pairs dealt to %d rank(s), "
                       "a query's hits merged and finished on its owner rank (q mod N), counters all-reduced, tables gathered on rank 0 (strong scaling)" % (total_mb, world),
           "n_gpus": world, "scaling": "strong", "items": len(items), "items_per_rank": [owner.count(r) for r in range(world)], "worker_contexts_per_rank": nwk,
           "ms_per_database_pass": dt * 1e3, "rank_busy_ms": busy_all, "rank0_gather_ms": t_gather / steps * 1e3, "rank0_finish_ms": t_finish / steps * 1e3,
           "queries_finished_on": "their owner rank (q mod N); the finished tables travel to rank 0", "queries_owned_by_rank0": len(my_q),
           "residues_per_s": sum(m["nres"] for m in merged) / dt, "hits": int(sum(t[0] for t in tables)), "hits_per_query": [t[0] for t in tables],
           "tables_sha1": hashlib.sha1("".join(t[1] for t in tables).encode()).hexdigest()[:16]}
    if on_gpu and world > 1:
        # the same job on rank 0 alone (outside the timed region): the N-rank search must print the single-rank search's tables
        solo = []
        for q, hmm in enumerate(hmms):
            om = ba.OProfile(ctx, ba.Profile(hmm))
            pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
            blk = ba.SeqBlock(ctx, [g[s_:s_ + n] for _, s_, n, _ in all_wins[q]]); blk.set_context([c for _, _, _, c in all_wins[q]])
            st, dm, _ = pipe.run_hits(blk)
            solo.append(finish_query(ba, hmm, dm, all_wins[q], st.nres, n_nt))
        out["tables_equal_to_single_rank_search"] = [a[1] for a in solo] == [b[1] for b in tables]
        out["single_rank_hits_per_query"] = [a[0] for a in solo]
        out["queries_whose_tables_differ"] = [q for q, (a, b) in enumerate(zip(solo, tables)) if a[1] != b[1]]
    return out


def c5_leg_ranks(ba, synth, bdist, ctx, args, rank, world, dev, on_gpu, sync, total_mb):
    """BASELINE configs[4] over N ranks: the windows of ONE <total_mb> Mb genome in contiguous shards (what the reference's block
    queue does), the whole --fs path per rank, counters reduced and domains gathered on rank 0 (strong scaling).  The pass lasts
    at least as long as the longest DNA window's row chains whatever N is: c5.size_sweep puts that floor on record."""
    n_nt = int(total_mb * 1e6)
    hmm = ba.HMM(c5_model_path())
    wins = bdist.split_targets([n_nt], hmm.max_length)
    lo, hi = bdist.shard_range(len(wins), rank, world)
    if on_gpu:
        g, planted = synth.genome(n_nt, seed=4400, hmms=[hmm], genes_per_model=max(8, n_nt // 400_000), frameshift=True)
        om = ba.OProfile(ctx, ba.Profile(hmm))
        om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
        om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
        pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
        blk = ba.SeqBlock(ctx, [g[s_:s_ + n] for _, s_, n, _ in wins[lo:hi]]); blk.set_context([c for _, _, _, c in wins[lo:hi]])
        del g
        before = 2 * sum(n - c for _, _, n, c in wins[:lo])               # what the search counted before this rank's shard (both strands)
        pipe.run_frameshift_domains(om3, om5, blk, nres_before=before)
    steps = 2
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        if on_gpu:
            stats, fw, dm, _ = pipe.run_frameshift_domains(om3, om5, blk, nres_before=before)
        else:
            stats = ba.PipelineStats(); stats.nres = 2 * sum(n - c for _, _, n, c in wins[lo:hi])
            dm = []
            for w in range(min(hi - lo, 2)):
                d = ba.FsDomain(); d.window = w; d.reported = 1; d.cigar = "%dM" % (w + 1)
                dm.append(d)
        gathered = bdist.gather_domains(dm, lo, 0, dev)
        merged = bdist.reduce_stats(stats, dev)
    sync()
    dt = bdist.max_over_ranks((time.perf_counter() - t0) / steps, dev)
    if rank != 0:
        return None
    out = {"workload": "synthetic %d-node model --fs vs ONE %.0f Mb synthetic genome, %d windows of %d nt with context in contiguous shards over %d rank(s); "
                       "domains gathered on rank 0 inside the timed region (strong scaling)" % (C5_M, total_mb, len(wins), bdist.BLOCK_LENGTH, world),
           "n_gpus": world, "scaling": "strong", "ms_per_pass": dt * 1e3, "residues_per_s": merged["nres"] / dt, "mode": "strict (the library's default)",
           "domains_gathered": len(gathered), "reported": int(sum(d.reported for d in gathered)),
           "windows_of_gathered_domains_are_global": bool(all(0 <= d.window < len(wins) for d in gathered))}
    if on_gpu and world > 1:
        # the same search on rank 0 alone, outside the timed region (bathsearch.c:868-921: what the merge of the workers must give):
        # every gathered domain and every counter equal to the single-rank pass's
        del blk
        g, _ = synth.genome(n_nt, seed=4400, hmms=[hmm], genes_per_model=max(8, n_nt // 400_000), frameshift=True)
        whole = ba.SeqBlock(ctx, [g[s_:s_ + n] for _, s_, n, _ in wins]); whole.set_context([c for _, _, _, c in wins])
        del g
        s_stats, _, s_dm, _ = pipe.run_frameshift_domains(om3, om5, whole)
        same, diff = records_diff(domain_records(gathered), domain_records(s_dm))
        cdiff = {f: (merged[f], int(getattr(s_stats, f))) for f in COUNTERS if merged[f] != int(getattr(s_stats, f))}
        out["domains_equal_to_single_rank_search"] = bool(same)
        out["counters_equal_to_single_rank_search"] = not cdiff
        out["single_rank_check"] = {**diff, "counter_mismatches": {k: {"ranks": a, "single": b} for k, (a, b) in cdiff.items()}}
    return out


def c5_size_sweep(ba, synth, bdist, ctx, sizes_mb):
    """configs[4] on ONE GPU at growing genome sizes: the pass is bound by the row chains of its longest DNA windows (one or two
    chain blocks per CU), so its time grows far slower than the genome until the windows outnumber the chip's chain slots --
    the floor that window sharding over N GPUs cannot get under."""
    hmm = ba.HMM(c5_model_path())
    om = ba.OProfile(ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    out = []
    for mb in sizes_mb:
        n_nt = int(mb * 1e6)
        g, planted = synth.genome(n_nt, seed=4400, hmms=[hmm], genes_per_model=max(8, n_nt // 400_000), frameshift=True)
        wins = bdist.split_targets([n_nt], hmm.max_length)
        blk = ba.SeqBlock(ctx, [g[s_:s_ + n] for _, s_, n, _ in wins]); blk.set_context([c for _, _, _, c in wins])
        del g
        pipe.run_frameshift_domains(om3, om5, blk, arrays=True)
        ctx.synchronize()
        t0 = time.perf_counter()
        stats, fw, dm, _ = pipe.run_frameshift_domains(om3, om5, blk, arrays=True)
        dt = time.perf_counter() - t0
        kt = pipe.kernel_times()
        out.append({"genome_mb": mb, "ms_per_pass": dt * 1e3, "residues_per_s": stats.nres / dt, "fs_windows": int(len(fw)),
                    "longest_fs_window_nt": int(fw["length"].max()) if len(fw) else 0, "domains": int(len(dm)), "planted": len(planted),
                    "chain_kernels_ms": {k: round(kt[k][0], 1) for k in ("fs3_fwd_kernel", "fs_bwd_kernel<3>") if k in kt}})
        del blk
    return out


GPU = 0          # the device this rank's contexts live on (main() sets it: LOCAL_RANK, or 0 under BATH_BENCH_SHARE_DEVICE=1)


# ------------------------------------------------------------------------------------------------ launcher

def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """`bench.py --gpus N` outside torch.distributed.run: start N ranks in a fresh child and relay rank 0's JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


# ------------------------------------------------------------------------------------------------ main

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=1_000_000, help="DNA windows per GPU (weak) or in total (strong); BASELINE config: 10^6")
    ap.add_argument("--length", type=int, default=1000)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fs", action="store_true", help="skip the configs[2] (--fs) leg")
    ap.add_argument("--fs-windows", type=int, default=1_000_000)
    ap.add_argument("--fs-parity-windows", type=int, default=20000, help="windows of the fs block the oracle's --fs pipeline scores before GPU init (fs.parity_check)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the host-fed (PCIe-inclusive) leg")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the leg with three worker contexts running blocks concurrently")
    ap.add_argument("--no-c45", action="store_true", help="skip the configs[3] / configs[4] legs (multi-HMM database; 1024-node model with --fs)")
    ap.add_argument("--c4-mb", type=float, default=12.5, help="genome of the configs[3] leg, Mb (100 Mb over 8 GPUs)")
    ap.add_argument("--c5-mb", type=float, default=125.0, help="genome of the configs[4] leg, Mb (1 Gb over 8 GPUs)")
    ap.add_argument("--c45-sample-windows", type=int, default=6, help="windows per model the CPU pipeline scores for the c4 parity check")
    ap.add_argument("--c5-sample-windows", type=int, default=2, help="genome windows the oracle's --fs pipeline scores for the c5 parity check")
    ap.add_argument("--c4-total-mb", type=float, default=100.0, help="genome of configs[3] as ONE job over the ranks (strong scaling; at N=1: c4.full_job)")
    ap.add_argument("--c5-total-mb", type=float, default=1000.0, help="genome of configs[4] as ONE job over the ranks (strong scaling, N>1)")
    ap.add_argument("--c5-sweep", type=str, default="125,250,500,1000", help="N=1: genome sizes (Mb) of c5.size_sweep; empty to skip")
    ap.add_argument("--no-one-part", action="store_true",
                    help="skip the whole-block-as-one-part passes behind roofline.valu (profiling: every ssv_orf_kernel launch is then a timed-step launch)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="CPU test hook: launcher, broadcast, reduce and gather over gloo with fabricated counters; no kernels, value 0")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import bath_amd as ba
    from bath_amd import synth

    # this rank's windows: weak = its own block of --windows; strong = its contiguous shard of one block of --windows
    from bath_amd.dist import shard_range
    if args.scaling == "strong":
        lo, hi = shard_range(args.windows, rank, world)
    else:
        lo, hi = rank * args.windows, (rank + 1) * args.windows
    n_mine = hi - lo

    # ---- before any GPU initialisation: the model, the synthetic block, the CPU baseline (forks workers)
    hmm0 = ba.HMM(MODEL)
    flat = offsets = None
    base = cpu_counters = c45_cpu = fs_data = fs_cpu = None
    cpu_covered = 0
    if not args.plumbing_only:
        if args.scaling == "strong":                      # one block for the whole job: every rank generates it and keeps its shard
            full, _, _ = synth.dna_windows(args.windows, args.length, seed=42, hmm=hmm0)
            flat = full[lo * args.length:hi * args.length].copy()
            del full
        else:
            flat, _, _ = synth.dna_windows(n_mine, args.length, seed=42 + rank, hmm=hmm0)
        offsets = np.arange(n_mine + 1, dtype=np.int64) * args.length
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            base, cpu_counters, cpu_covered = cpu_baseline(flat, args.length, n_mine)
            if not args.no_c45:
                c45_cpu = c45_cpu_samples(args)
        if rank == 0 and world == 1 and not args.no_fs:
            fs_data = synth.dna_windows(args.fs_windows, args.length, seed=4242, hmm=hmm0, frameshift=True)
            if not args.no_cpu_baseline and args.fs_parity_windows > 0:
                fs_cpu = fs_cpu_sample(fs_data[0], args.length, min(args.fs_parity_windows, args.fs_windows))
                nw = min(args.fs_windows, 400_000)
                fs_cpu["baseline"] = fs_cpu_baseline(MODEL, [fs_data[0][i * args.length:(i + 1) * args.length] for i in range(nw)], None,
                                                     "windows x %d nt of the fs block (both strands)" % args.length, probe=2000)

    import torch
    import torch.distributed as dist
    from bath_amd import dist as bdist

    on_gpu = torch.cuda.is_available() and not args.plumbing_only
    if not on_gpu and not args.plumbing_only:
        raise SystemExit("bench.py needs an MI355X: the backend has no CPU path")
    # Which GPU a rank computes on, and where the collectives' tensors live.  Default: rank r on GPU r, RCCL ("nccl") on device tensors.
    # BATH_BENCH_SHARE_DEVICE=1 puts every rank's context on GPU 0 and BATH_BENCH_BACKEND=gloo moves the collectives to CPU tensors:
    # the N-rank legs then run with REAL kernels on a one-GPU box (RCCL refuses two ranks on one device), which is how
    # tests/test_nrank_gpu.py checks the N-rank search against the single-rank search record for record.
    global GPU
    share_device = os.environ.get("BATH_BENCH_SHARE_DEVICE") == "1"
    backend = os.environ.get("BATH_BENCH_BACKEND") or ("nccl" if on_gpu else "gloo")
    if backend not in ("nccl", "gloo"):
        raise SystemExit("BATH_BENCH_BACKEND must be nccl or gloo")
    if on_gpu and share_device and backend == "nccl" and world > 1:
        raise SystemExit("BATH_BENCH_SHARE_DEVICE=1 needs BATH_BENCH_BACKEND=gloo: RCCL does not run two ranks on one device")
    GPU = 0 if share_device else local_rank
    if on_gpu:
        torch.cuda.set_device(GPU)
    dev = torch.device("cuda", GPU) if (on_gpu and backend == "nccl") else torch.device("cpu")
    if world > 1:
        if backend == "nccl" and on_gpu:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    # query model: read on rank 0, broadcast (RCCL over xGMI), parsed by every rank
    blob = open(MODEL, "rb").read() if rank == 0 else b""
    blob = bdist.broadcast_bytes(blob, 0, dev)
    tmp = "/tmp/bath_bench_model_%d.bhmm" % os.getpid()
    with open(tmp, "wb") as fh:
        fh.write(blob)
    hmm = ba.HMM(tmp)
    os.unlink(tmp)

    def sync():
        if on_gpu:
            ctx.synchronize()
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()

    stage_ms, stage_launches = {}, {}
    if on_gpu:
        ctx = ba.Context(GPU)
        om = ba.OProfile(ctx, ba.Profile(hmm))
        dna = ba.SeqBlock(ctx, flat, offsets)
        pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
        stats = None
        # configs[3]'s leg runs FIRST, before this process has created the lanes, side streams and helper contexts of the other legs: its
        # six worker contexts launch thousands of ~10 us kernels per pass, and how HIP maps their streams onto the hardware queues depends
        # on what the process created before (measured: 18.7 ms per database pass here, 26.5 after the cascade leg, 33 after the streamed
        # and concurrent-blocks legs; the cascade and --fs legs are insensitive to the order).  BATH_BENCH_C4_LAST=1 restores the old order.
        c4_early = None
        if world == 1 and not args.no_c45 and os.environ.get("BATH_BENCH_C4_LAST") != "1":
            c4_early = c4_leg(ba, synth, bdist, ctx, args, c45_cpu)
            c4_early["full_job"] = c4_leg_ranks(ba, synth, bdist, ctx, args, 0, 1, dev, True, sync, args.c4_total_mb)     # the N = 1 point of the N-rank leg
        for _ in range(args.warmup):
            stats, _ = pipe.run(dna, want_results=False)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            stats, _ = pipe.run(dna, want_results=False)
            for name, ms, nl in pipe.timings():
                stage_ms.setdefault(name, []).append(ms)
                stage_launches[name] = nl
        sync()
        elapsed = bdist.max_over_ranks(time.perf_counter() - t0, dev)
        # one more pass with the copy-out, to gather the records on rank 0 (outside the timed region; timed on its own below)
        pipe.run(dna, want_results=True, copy=False)           # first call sizes the page-locked result array
        t1 = time.perf_counter()
        stats, res = pipe.run(dna, want_results=True, copy=False)
        ms_with_results = (time.perf_counter() - t1) * 1e3
        res = res.copy()
    else:                                                  # --plumbing-only: fabricated counters, the collectives are what runs
        elapsed = bdist.max_over_ranks(1e-3, dev)
        stats = ba.PipelineStats()
        stats.nres, stats.n_orfs, stats.n_past_msv = 2 * n_mine * args.length, 38 * n_mine, n_mine
        res = np.zeros(min(n_mine, 64), dtype=ba.ORF_RESULT_DTYPE)
        res["window"] = np.arange(len(res))
        ms_with_results = 0.0
    merged = bdist.reduce_stats(stats, dev)
    hits = bdist.gather_results(res, lo, 0, dev)
    strong_check = None
    if on_gpu and world > 1 and args.scaling == "strong" and rank == 0:
        # one block over N ranks must be the block on one rank (p7_pipeline_Merge, p7_tophits_Merge: bathsearch.c:884-893): rank 0 runs
        # the WHOLE block alone, outside the timed region, and compares the reduced counters and the gathered records with its own
        full, _, _ = synth.dna_windows(args.windows, args.length, seed=42, hmm=hmm0)
        whole = ba.SeqBlock(ctx, full, np.arange(args.windows + 1, dtype=np.int64) * args.length)
        s_stats, s_res = pipe.run(whole, want_results=True, copy=False)
        cdiff = {f: (merged[f], int(getattr(s_stats, f))) for f in COUNTERS if merged[f] != int(getattr(s_stats, f))}
        key = lambda r: sorted(zip(*([r[f].tolist() for f in ("window", "strand", "frame", "start", "end", "stage")] + [np.ascontiguousarray(r["usc"]).view(np.uint32).tolist()])))
        strong_check = {"counters_equal_to_single_rank": not cdiff, "records_equal_to_single_rank": key(hits) == key(s_res),
                        "counter_mismatches": {k: {"ranks": a, "single": b} for k, (a, b) in cdiff.items()}}
        del whole, full

    out = None
    if rank == 0:
        tot = merged
        nres_step = tot["nres"]
        cells_step = tot["cells_msv"] + tot["cells_vit"] + tot["cells_fwd"]
        ms_step = elapsed / args.steps * 1e3
        value = nres_step / (elapsed / args.steps) if on_gpu else 0.0
        out = {
            "metric": "DP Gcells/s + residues/s through bathsearch pipeline at 1/2/4/8 MI355X",
            "value": value, "unit": "residues/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "integer scores: u8 as exact binary16 (SSV), u8 (MSV), i16 (Viterbi); f32 (Forward, bias)", "data": "synthetic",
            "config": {"workload": "Caudal_act.bhmm (M=%d) vs %d x %d nt iid DNA windows %s (1%% planted), both strands, "
                                   "6-frame translation + MSV/bias/Viterbi/Forward filter cascade, codon table %d, no --fs"
                                   % (hmm.M, args.windows, args.length, "per GPU" if args.scaling == "weak" else "in total, sharded over the GPUs", hmm.ct),
                       "windows_per_gpu": n_mine, "window_nt": args.length, "M": hmm.M},
            "gcells_per_s": cells_step / (elapsed / args.steps) / 1e9 if on_gpu else 0.0,
            "cells_per_step": cells_step, "residues_per_step": nres_step,
            "survivors": {k: tot[k] for k in ("n_orfs", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd")},
            "hits_gathered": int(len(hits)) if hits is not None else 0,
            "timed_region": "DNA resident in HBM (1 byte per nucleotide); counters only -- the surviving ORFs' records stay on the device; "
                            "a pass that also copies them to the host takes ms_per_step_with_records",
            "ms_per_step_with_records": ms_with_results,
        }
        if args.plumbing_only:
            out["plumbing_only"] = True
        if strong_check is not None:
            out["strong_scaling_check"] = strong_check
    if rank == 0 and on_gpu:
        out["stage_ms"] = {k: float(np.mean(v)) for k, v in stage_ms.items()}
        # dominant kernel: ssv_orf_kernel.  A large block runs as <lanes> concurrent parts (bath_hip_pipeline_filters), so a
        # step launches the kernel <lanes> times, each on its part of the ORF list and overlapping the other parts' work.
        lanes = max(1, int(stage_launches.get("ssv_f1", 1)))
        k_ms = float(np.mean(stage_ms["ssv_f1"])) / lanes                 # average duration of one launch (HIP events, its own stream)
        orf_res = stats.cells_msv / hmm.M
        # 1 B per ORF residue + 16 B per ORF work-list record read once, ~34 B written per surviving ORF; per launch
        algo_bytes = (orf_res + 16.0 * stats.n_orfs + 34.0 * stats.n_past_msv) / lanes
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        # the same kernel alone on the chip: one part, nothing overlapping it (outside the timed region)
        one = []
        if not args.no_one_part:
            os.environ["BATH_HIP_LANES"] = "1"
            pipe.run(dna, want_results=False)
            for _ in range(3):
                pipe.run(dna, want_results=False)
                one.append({n: ms for n, ms, _ in pipe.timings()})
            del os.environ["BATH_HIP_LANES"]
        k1_ms = float(np.mean([o["ssv_f1"] for o in one])) if one else float("nan")
        tc1 = stats.cells_msv / (k1_ms * 1e-3) / 1e12 if one else float("nan")
        traffic = None
        pmc = next((f for f in (os.path.join(ROOT, "profiles", r + "_ssv_orf_pmc.json") for r in ("r06", "r05", "r04", "r03", "r02")) if os.path.exists(f)), "")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch_full_block") / lanes
            except Exception:
                traffic = None
        out["roofline"] = {
            # "bound" names what binds the kernel: the issue rate of packed 16-bit VALU ops (the "valu" object below: 0.75 of the measured
            # ceiling).  achieved / peak / frac stay the HBM figures the contract asks for -- a few percent by construction (0.01 B per DP cell)
            "bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "binding_resource": "valu (packed 16-bit issue rate; the fraction of THAT ceiling is valu.frac); achieved / peak / frac / traffic are the HBM figures",
            "traffic": traffic, "traffic_source": "rocprofv3 --pmc passes of this command, profiles/%s (static: counters cannot be read from inside the run)" % os.path.basename(pmc),
            "kernel": "ssv_orf_kernel", "kernel_ms": k_ms, "launches_per_step": lanes,
            "note": "DP rows held in VGPRs (integer scores as binary16): compulsory HBM traffic is 1 B per ORF residue, so the HBM fraction is small by "
                    "construction; the kernel is bound by the issue rate of packed 16-bit VALU ops (DESIGN.md 4.1), see valu",
            # the bound that does apply: tools/valu_rate.hip measures 5.2e11 wave-instructions/s for v_pk_add_f16 / v_pk_maximum3_f16 at 8 waves per
            # SIMD (4.4e11 at the 4 this kernel's registers allow); a row costs 3 of them per 4 cells, 64 lanes each: 44.4 (37.9) Tcells/s
            "valu": {"kernel_ms_one_part": k1_ms, "tcells_per_s": tc1, "peak_tcells_per_s": 44.4, "frac": tc1 / 44.4,
                     "peak_at_this_occupancy_tcells_per_s": 37.9, "frac_at_this_occupancy": tc1 / 37.9,
                     "how": "whole block as ONE part (BATH_HIP_LANES=1), nothing overlapping the kernel; HIP events on its stream",
                     "peak_source": "measured packed-op issue rate, tools/valu_rate.hip, profiles/r01_valu_rate.txt"},
            "one_part_stage_ms": {k: float(np.mean([o[k] for o in one])) for k in one[0]} if one else None,
        }
        if not one:
            out["roofline"]["valu"] = None
        else:
            out["one_part_kernel_sum_ms"] = float(sum(out["roofline"]["one_part_stage_ms"].values()))
        if base is not None:
            out["cpu_baseline"] = base
            # the CPU leg scored windows [0, cpu_covered) of this very block: compare every counter with the GPU's over the same windows
            if cpu_covered == n_mine:
                g = {f: int(getattr(stats, f)) for f in COUNTERS}
            else:
                sub = ba.SeqBlock(ctx, flat[:cpu_covered * args.length], offsets[:cpu_covered + 1])
                gs, _ = pipe.run(sub, want_results=False)
                g = {f: int(getattr(gs, f)) for f in COUNTERS}
            diff = {f: (g[f], cpu_counters[f]) for f in COUNTERS if g[f] != cpu_counters[f]}
            out["parity_full_block"] = (not diff) and cpu_covered == n_mine
            out["parity_check"] = {"windows": cpu_covered, "of": n_mine, "counters": list(COUNTERS), "all_equal": not diff,
                                   "mismatches": {k: {"gpu": v[0], "cpu": v[1]} for k, v in diff.items()}}
        if world == 1 and not args.no_streamed:
            out.update(streamed_leg(ba, ctx, pipe, flat, offsets, args, stats))
        if world == 1 and not args.no_concurrent:
            out["concurrent_blocks"] = concurrent_leg(ba, hmm, flat, offsets, args, stats, ms_step)
        if not args.no_fs and world == 1:
            out["fs"] = fs_leg(ba, synth, ctx, hmm, om, args, fs_data, fs_cpu)
        if not args.no_c45 and world == 1:
            if c4_early is not None:
                out["c4"] = c4_early
            else:
                out["c4"] = c4_leg(ba, synth, bdist, ctx, args, c45_cpu)
                out["c4"]["full_job"] = c4_leg_ranks(ba, synth, bdist, ctx, args, 0, 1, dev, True, sync, args.c4_total_mb)
            out["c5"] = c5_leg(ba, synth, bdist, ctx, args, c45_cpu)
            if args.c5_sweep:
                out["c5"]["size_sweep"] = c5_size_sweep(ba, synth, bdist, ctx, [float(x) for x in args.c5_sweep.split(",")])
    if world > 1 and not args.no_c45:                       # configs[3] / configs[4] as one job over the ranks; rank 0 reports
        c4_multi = c4_leg_ranks(ba, synth, bdist, ctx if on_gpu else None, args, rank, world, dev, on_gpu, sync, args.c4_total_mb)
        c5_multi = c5_leg_ranks(ba, synth, bdist, ctx if on_gpu else None, args, rank, world, dev, on_gpu, sync, args.c5_total_mb)
        if rank == 0:
            out["c4"], out["c5"] = c4_multi, c5_multi
    if world > 1 and not args.no_fs:                        # every rank takes part; rank 0 reports
        fs_multi = fs_leg_ranks(ba, synth, bdist, dist, ctx if on_gpu else None, hmm, om if on_gpu else None, args, rank, world, dev, on_gpu, sync)
        if rank == 0:
            out["fs"] = fs_multi
    if rank == 0:
        # The driver keeps the LAST 2000 characters of the output: the full record (15-20 KB with the per-model arrays and the notes) goes
        # to stderr and to bench_detail.json, and stdout ends with one compact line that carries the contract's keys and every leg's numbers.
        full = json.dumps(out)
        sys.stderr.write(full + "\n")
        sys.stderr.flush()
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as fh:
                fh.write(full + "\n")
        except OSError:
            pass
        print(fit_line(compact_line(out)))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _r(x, sig=4):
    """A number rounded to <sig> significant digits (None stays None)."""
    if x is None or isinstance(x, bool) or not isinstance(x, (int, float)):
        return x
    if x == 0 or x != x or x in (float("inf"), float("-inf")):
        return x
    return float("%.*g" % (sig, x)) if isinstance(x, float) else x


def compact_line(out):
    """The bench line the driver records (its tail holds 2000 characters): the contract's keys with short strings, and per leg the
    numbers the reviews quote -- fs (configs[2]), c4 (configs[3]), c5 (configs[4]).  Everything else is in the full record (stderr,
    gpurun_out/bench_detail.json)."""
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")}
    c["value"], c["ms_per_step"] = _r(c["value"], 5), _r(c["ms_per_step"], 5)
    c["dtype"] = "u8/i16 integer scores (SSV: exact binary16), f32 Forward"
    c["data"] = out.get("data")
    cfg = out.get("config", {})
    c["config"] = {"workload": "Caudal_act.bhmm M=%s vs %s x %s nt DNA windows/GPU, both strands, 6-frame + MSV/bias/Vit/Fwd cascade, no --fs"
                               % (cfg.get("M"), cfg.get("windows_per_gpu"), cfg.get("window_nt"))}
    c["gcells_per_s"] = _r(out.get("gcells_per_s"))
    if out.get("plumbing_only"):
        c["plumbing_only"] = True
    rf = out.get("roofline")
    if rf:
        c["roofline"] = {"bound": rf["bound"], "achieved": _r(rf["achieved"]), "peak": rf["peak"], "unit": rf["unit"], "frac": _r(rf["frac"]),
                         "traffic": _r(rf.get("traffic")), "kernel": rf.get("kernel"), "kernel_ms": _r(rf.get("kernel_ms")),
                         "valu": {"achieved": _r(g(rf, "valu", "tcells_per_s")), "peak": g(rf, "valu", "peak_tcells_per_s"), "unit": "Tcells/s",
                                  "frac": _r(g(rf, "valu", "frac"))}}
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "sample": "%s (SSE2 restatement of impl_sse)" % cb.get("sample", "").split(",")[0]}
    if "strong_scaling_check" in out:
        c["strong_check"] = {k: out["strong_scaling_check"][k] for k in ("counters_equal_to_single_rank", "records_equal_to_single_rank")}
    if "parity_full_block" in out:
        c["parity_full_block"] = out["parity_full_block"]
    fs = out.get("fs")
    if fs:
        pc = fs.get("parity_check") or {}
        c["fs"] = {"ms_per_pass": _r(fs.get("ms_per_pass")), "residues_per_s": _r(fs.get("residues_per_s")),
                   "roofline": {"frac": _r(g(fs, "roofline", "frac")), "bytes_per_cell": _r(g(fs, "roofline", "bytes_per_cell")),
                                "alone_frac": _r(g(fs, "roofline", "alone", "frac")), "alone_ms": _r(g(fs, "roofline", "alone", "ms"))},
                   "parity_ok": bool(pc) and all(v is True for k, v in pc.items() if isinstance(v, bool)) if pc else None,
                   "two_workers_ms_per_block": _r(g(fs, "concurrent_blocks", "ms_per_block")),
                   "fast_ms": _r(g(fs, "fast", "ms_per_pass")), "domains": fs.get("domains"),
                   "fast_identical": g(fs, "fast", "domains_identical_to_strict_mode")}
        if fs.get("cpu_baseline"):
            c["fs"]["cpu_baseline"] = {"value": _r(g(fs, "cpu_baseline", "value")), "cores": g(fs, "cpu_baseline", "cores"), "kind": "port: SSE2 cascade + SSE2 fs3 parsers + scalar fs5 envelopes"}
        if fs.get("n_gpus"):
            c["fs"] = {"ms_per_pass": _r(fs.get("ms_per_pass")), "residues_per_s": _r(fs.get("residues_per_s")), "n_gpus": fs.get("n_gpus"),
                       "domains_equal": fs.get("domains_equal_to_single_rank_search"), "counters_equal": fs.get("counters_equal_to_single_rank_search")}
    c4 = out.get("c4")
    if c4:
        c["c4"] = {"ms_per_database_pass": _r(c4.get("ms_per_database_pass")), "concurrent_queries_ms": _r(g(c4, "concurrent_queries", "ms_per_database_pass")),
                   "full_job_ms": _r(g(c4, "full_job", "ms_per_database_pass")) if "full_job" in c4 else None,
                   "hits_equal": g(c4, "concurrent_queries", "hits_equal_to_serial_loop"), "parity_ok": g(c4, "parity_check", "all_equal")}
        if c4.get("n_gpus"):
            c["c4"].update({"n_gpus": c4.get("n_gpus"), "tables_equal": c4.get("tables_equal_to_single_rank_search")})
            for k in ("concurrent_queries_ms", "full_job_ms", "hits_equal", "parity_ok"):       # the N = 1 leg's keys
                if c["c4"].get(k) is None:
                    c["c4"].pop(k, None)
    c5 = out.get("c5")
    if c5:
        sw = c5.get("size_sweep") or []
        c["c5"] = {"ms_per_pass": _r(c5.get("ms_per_pass")), "fast_ms": _r(g(c5, "fast", "ms_per_pass")),
                   "sweep_mb_ms": [[int(x["genome_mb"]), _r(x["ms_per_pass"])] for x in sw], "parity_ok": g(c5, "parity_check", "all_equal")}
        if c5.get("cpu_baseline"):
            c["c5"]["cpu_baseline"] = {"value": _r(g(c5, "cpu_baseline", "value")), "cores": g(c5, "cpu_baseline", "cores")}
        if c5.get("n_gpus"):
            c["c5"] = {"ms_per_pass": _r(c5.get("ms_per_pass")), "n_gpus": c5.get("n_gpus"), "domains_equal": c5.get("domains_equal_to_single_rank_search"),
                       "counters_equal": c5.get("counters_equal_to_single_rank_search")}
    c["detail"] = "full record: stderr + gpurun_out/bench_detail.json"
    return c


def fit_line(c, limit=2000):
    """The compact record as one JSON line under <limit> characters: when it does not fit, optional keys go, least important first,
    and the contract's keys (metric ... config, roofline, cpu_baseline) never do -- a long line must not cost the run its record."""
    dumps = lambda d: json.dumps(d, separators=(",", ":"))
    line = dumps(c)
    optional = [("detail",), ("c5", "sweep_mb_ms"), ("fs", "fast_identical"), ("fs", "domains"), ("c4", "hits_equal"), ("fs", "fast_ms"), ("c5", "fast_ms"),
                ("fs", "roofline"), ("gcells_per_s",), ("parity_full_block",), ("c5",), ("c4",), ("fs",)]
    dropped = []
    for path in optional:
        if len(line) < limit:
            break
        d = c
        for k in path[:-1]:
            d = d.get(k) if isinstance(d, dict) else None
        if isinstance(d, dict) and path[-1] in d:
            del d[path[-1]]
            dropped.append(".".join(path))
            c["dropped_to_fit"] = dropped
            line = dumps(c)
    return line


def concurrent_leg(ba, hmm, flat, offsets, args, stats_one, ms_one, workers=3, main_ctx=None):
    """The reference's threading model on one GPU (bathsearch.c thread_loop: every worker thread owns a block and a pipeline
    object): <workers> contexts, each running the cascade over its own resident block at the same time, so that one block's
    translation (head) and survivor kernels (tail) run beside another block's SSV.  Outside the headline's timed region."""
    import threading
    objs = []
    if main_ctx is not None:
        main_ctx.trim()                                         # (see fs_concurrent_leg)
    for _ in range(workers):
        c = ba.Context(GPU)
        o = ba.OProfile(c, ba.Profile(hmm))
        d = ba.SeqBlock(c, flat, offsets)
        p = ba.Pipeline(c, o, fs_pipe=False, ncbi_table=hmm.ct)
        p.run(d, want_results=False)
        objs.append((c, o, d, p))
    per = max(2, args.steps)
    got = [None] * workers

    def work(w):
        c, _, d, p = objs[w]
        for _ in range(per):
            got[w], _ = p.run(d, want_results=False)
        c.synchronize()

    for c, _, _, _ in objs:
        c.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(w,)) for w in range(workers)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    same = all(int(getattr(g, f)) == int(getattr(stats_one, f)) for g in got for f in COUNTERS)
    ms = dt * 1e3 / (per * workers)
    for c, _, _, _ in objs:
        c.close()
    return {"what": "%d worker contexts, each a block of the headline's size resident in HBM and its own pipeline object, cascades running "
                    "concurrently (heads and tails of one block beside the SSV of another)" % workers,
            "workers": workers, "blocks": per * workers, "ms_per_block": ms, "residues_per_s": stats_one.nres / (ms * 1e-3),
            "vs_one_worker": ms_one / ms, "counters_equal_to_one_worker": bool(same),
            "note": "the gain is bounded by the SSV kernel, which saturates the packed-16-bit VALU issue rate whatever runs beside it (roofline.valu)"}


def streamed_leg(ba, ctx, pipe, flat, offsets, args, stats_resident):
    """The same block fed from the host every step (what a search over more DNA than fits looks like): the block arrives in 2 bits
    per nucleotide from page-locked memory on a copy stream, the upload of block k+1 overlapping the cascade of block k
    (two device blocks alternate), a kernel expands it on the device; the surviving ORFs' records are copied to the host
    inside the step.  PCIe-inclusive: reported next to `value`, never as `value`."""
    packed, es, ep, ec = ba.pack2(flat, offsets)               # the input format of this mode; packing is the reader's job
    pins = [ba.PinnedBuffer(len(packed)) for _ in range(2)]
    for p_ in pins:
        p_.array[:len(packed)] = packed
    blocks = [ba.StreamedBlock(ctx, offsets) for _ in range(2)]
    t_up0 = time.perf_counter()
    blocks[0].upload(pins[0], es, ep, ec)
    blocks[0].wait()
    ctx.synchronize()
    upload_ms = (time.perf_counter() - t_up0) * 1e3           # one upload + expansion with nothing to hide behind
    blocks[0].upload(pins[0], es, ep, ec)
    steps = max(4, args.steps)
    out = {}
    for with_records in (False, True):
        for s in range(2):                                     # warm-up: both blocks through the cascade once
            blocks[1 - s % 2].upload(pins[1 - s % 2], es, ep, ec)
            blocks[s % 2].wait()
            pipe.run(blocks[s % 2], want_results=with_records)
        ctx.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            cur = s % 2
            blocks[1 - cur].upload(pins[1 - cur], es, ep, ec)  # next block: copy stream, asynchronous
            blocks[cur].wait()
            st, res = pipe.run(blocks[cur], want_results=with_records, copy=False)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        assert st.n_past_fwd == stats_resident.n_past_fwd and st.pos_past_msv == stats_resident.pos_past_msv
        key = "with_records" if with_records else "counters_only"
        out[key] = {"ms_per_step": dt * 1e3, "residues_per_s": st.nres / dt}
    return {"value_streamed": out["with_records"]["residues_per_s"],
            "streamed": {"what": "block re-fed from the host every step: 2-bit packed DNA (%.0f MB per block instead of %.0f MB) from page-locked memory on a copy "
                                 "stream, upload of block k+1 overlapping the cascade of block k, expansion kernel on the device, surviving ORF records "
                                 "copied to the host inside the step (with_records)" % (len(packed) / 1e6, len(flat) / 1e6),
                         "steps": steps, "upload_and_expand_ms_unoverlapped": upload_ms, **out}}


def fs_leg_ranks(ba, synth, bdist, dist, ctx, hmm, om, args, rank, world, dev, on_gpu, sync):
    """BASELINE configs[2] under --gpus N: ONE block of --fs-windows windows dealt to the ranks in contiguous shards (strong scaling,
    what the reference's block queue does); every rank runs the whole --fs path on its shard, the counters are reduced and the
    domains -- records and CIGAR strings -- gathered on rank 0 (dist.gather_domains), inside the timed region."""
    lo, hi = bdist.shard_range(args.fs_windows, rank, world)
    if on_gpu:
        flat, offsets, _ = synth.dna_windows(args.fs_windows, args.length, seed=4242, hmm=hmm, frameshift=True)
        mine = flat[lo * args.length:hi * args.length].copy()
        del flat
        dna = ba.SeqBlock(ctx, mine, np.arange(hi - lo + 1, dtype=np.int64) * args.length)
        om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
        om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
        pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
        before = 2 * args.length * lo                                      # what the search counted before this rank's shard (both strands)
        pipe.run_frameshift_domains(om3, om5, dna, nres_before=before)
    steps = 2
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        if on_gpu:
            stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, dna, nres_before=before)
        else:                                              # --plumbing-only: fabricated records, the collectives are what runs
            stats = ba.PipelineStats(); stats.nres = 2 * (hi - lo) * args.length
            dm = []
            for w in range(min(hi - lo, 3)):
                d = ba.FsDomain(); d.window = w; d.reported = 1; d.cigar = "%dM" % (w + 1)
                dm.append(d)
        gathered = bdist.gather_domains(dm, lo, 0, dev)
        merged = bdist.reduce_stats(stats, dev)
    sync()
    dt = bdist.max_over_ranks((time.perf_counter() - t0) / steps, dev)
    if rank != 0:
        return None
    out = {"workload": "Caudal_act.bhmm --fs vs %d x %d nt windows in total, sharded over %d GPUs (strong scaling); domains gathered on rank 0 inside the timed region"
                       % (args.fs_windows, args.length, world),
           "ms_per_pass": dt * 1e3, "residues_per_s": merged["nres"] / dt, "scaling": "strong", "n_gpus": world,
           "domains_gathered": len(gathered), "windows_of_gathered_domains_are_global": bool(all(0 <= d.window < args.fs_windows for d in gathered)),
           "mode": "strict (the library's default)"}
    if on_gpu and world > 1:
        # the same block on rank 0 alone, outside the timed region: the merged search must be the single-rank search, record for record
        del dna
        flat, offsets, _ = synth.dna_windows(args.fs_windows, args.length, seed=4242, hmm=hmm, frameshift=True)
        whole = ba.SeqBlock(ctx, flat, offsets)
        s_stats, _, s_dm, _ = pipe.run_frameshift_domains(om3, om5, whole)
        same, diff = records_diff(domain_records(gathered), domain_records(s_dm))
        cdiff = {f: (merged[f], int(getattr(s_stats, f))) for f in COUNTERS if merged[f] != int(getattr(s_stats, f))}
        out["domains_equal_to_single_rank_search"] = bool(same)
        out["counters_equal_to_single_rank_search"] = not cdiff
        out["single_rank_check"] = {**diff, "counter_mismatches": {k: {"ranks": a, "single": b} for k, (a, b) in cdiff.items()}}
    return out


def fs_concurrent_leg(ba, hmm, flat, offsets, strict_keys, one_ms, workers=2, passes=6, main_ctx=None):
    """The --fs pass with <workers> worker contexts on one GPU, each owning a block and running whole strict passes on it at the same
    time (the reference's worker threads each own a block, bathsearch.c:1119-1290): one worker's cascade / envelope / host stages
    beside another's chain kernels.  Every worker's domains must be those of the one-worker pass."""
    import threading
    objs = []
    if main_ctx is not None:
        main_ctx.trim()                                         # the idle main context gives its lanes' and side contexts' streams back (hardware queues)
    for _ in range(workers):
        c = ba.Context(GPU)
        o = ba.OProfile(c, ba.Profile(hmm))
        o3 = ba.FSOProfile(c, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
        o5 = ba.FSOProfile(c, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
        d = ba.SeqBlock(c, flat, offsets)
        p = ba.Pipeline(c, o, fs_pipe=True, ncbi_table=hmm.ct)
        p.run_frameshift_domains(o3, o5, d, arrays=True)
        objs.append((c, o, o3, o5, d, p))
    for o in objs:
        o[0].synchronize()
    got = [None] * workers
    keys = ("window", "strand", "ienv", "jenv", "iali", "jali", "ihmm", "jhmm", "n_shifted_codons")

    def work(w):
        c, _, o3, o5, d, p = objs[w]
        for _ in range(passes):
            _, _, dm, _ = p.run_frameshift_domains(o3, o5, d, arrays=True)
        c.synchronize()
        got[w] = dm                                             # (a view of the library's records: compared after the clock stops)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(w,)) for w in range(workers)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    ms = (time.perf_counter() - t0) * 1e3 / (workers * passes)
    same = all({tuple(int(r[k]) for k in keys) + (fbits(r["envsc"]),) for r in g} == strict_keys for g in got)
    for o in objs:
        o[0].close()
    return {"what": "%d worker contexts, each a block of the fs leg's size resident in HBM with its own pipeline object, whole strict --fs passes running "
                    "concurrently" % workers,
            "workers": workers, "blocks": workers * passes, "ms_per_block": ms, "vs_one_worker": one_ms / ms,
            "domains_identical_to_one_worker_pass_incl_envsc_bits": bool(same)}


def fs_envelopes_alone(ba, ctx, om5, flat, offsets, fw, dm, M, reps=3):
    """The 5-codon envelope kernels with the chip to themselves: the envelopes of the pass's frameshift-branch domains as ONE batch through
    bath_hip_fs5_envelopes, Backward after Forward (bath_hip_set_fs_serial), nothing else running -- the counterpart of roofline.valu's
    one-part pass for the cascade.  In the pass the same kernels run side by side and beside the regions' Forward, in two batches."""
    sel = dm[fw["branch"][dm["fs_window"]] == 1]
    if len(sel) == 0:
        return None
    envs = []
    for r in sel:
        lo, hi = sorted((int(r["ienv"]), int(r["jenv"])))
        e = flat[int(offsets[int(r["window"])]) + lo - 1:int(offsets[int(r["window"])]) + hi]
        envs.append((3 - e[::-1]).astype(np.uint8) if int(r["strand"]) else e)
    blk = ba.SeqBlock(ctx, envs)

    def times():
        arr = (ba.KernelTime * 32)()
        n = ba.lib().bath_hip_kernel_times(ctx._h, 32, arr)
        return {arr[i].name.decode(): (float(arr[i].ms), float(arr[i].bytes)) for i in range(n)}

    ctx.set_fs_serial(True)
    try:
        ba.FS5Envelopes(ctx, om5, blk, logsum=ba.LOGSUM_TABLE_SERIAL)
        t0 = times()
        for _ in range(reps):
            ba.FS5Envelopes(ctx, om5, blk, logsum=ba.LOGSUM_TABLE_SERIAL)
        t1 = times()
    finally:
        ctx.set_fs_serial(None)
    names = [n for n in ("fs5_fwd_kernel", "fs_bwd_kernel<5>", "fs5_decode_oa_kernel") if n in t1]
    ms = {n: (t1[n][0] - t0.get(n, (0.0, 0.0))[0]) / reps for n in names}
    nbytes = {n: (t1[n][1] - t0.get(n, (0.0, 0.0))[1]) / reps for n in names}
    tot_ms, tot_b = sum(ms.values()), sum(nbytes.values())
    L = np.array([len(e) for e in envs], dtype=np.int64)
    return {"what": "the envelopes of the pass's frameshift-branch domains as one batch, Backward after Forward, nothing else on the chip (strict mode)",
            "envelopes": int(len(envs)), "cells": int(((L + 1) * (M + 1)).sum()), "kernel_ms": ms, "ms": tot_ms,
            "achieved": tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if tot_ms > 0 else None,
            "per_kernel_frac": {n: nbytes[n] / (ms[n] * 1e-3) / 1e9 / HBM_PEAK_GBS for n in names if ms[n] > 0}}


def fs_leg(ba, synth, ctx, hmm, om, args, data=None, cpu=None):
    """BASELINE configs[2]: --fs on a block whose planted domains are frameshifted."""
    flat, offsets, planted = data if data is not None else synth.dna_windows(args.fs_windows, args.length, seed=4242, hmm=hmm, frameshift=True)
    dna = ba.SeqBlock(ctx, flat, offsets)
    om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    parity = fs_parity_check(ba, ctx, pipe, om3, om5, flat, args.length, cpu) if cpu is not None else None
    pipe.run_frameshift_domains(om3, om5, dna, arrays=True)
    steps = 3
    kt = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, dna, arrays=True)     # record arrays: views of the library's memory
        for name, (ms, nl, cells, nbytes) in pipe.kernel_times().items():
            k = kt.setdefault(name, {"ms": 0.0, "launches": 0, "cells": 0.0, "bytes": 0.0})
            k["ms"] += ms / steps; k["launches"] += nl / steps; k["cells"] += cells / steps; k["bytes"] += nbytes / steps
    dt = (time.perf_counter() - t0) / steps
    for k in kt.values():
        k["gcells_per_s"] = k["cells"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
        k["algorithmic_GBps"] = k["bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
        k["hbm_frac"] = k["algorithmic_GBps"] / HBM_PEAK_GBS
    env = [n for n in ("fs5_fwd_kernel", "fs_bwd_kernel<5>", "fs5_decode_oa_kernel", "fs5_decode_kernel", "fs5_oa_kernel") if n in kt]
    env_ms = sum(kt[n]["ms"] for n in env)
    env_bytes = sum(kt[n]["bytes"] for n in env)
    env_cells = kt[env[0]]["cells"] if env else 0.0
    summary = {"fs_windows": int(len(fw)), "fs_window_nt": int(fw["length"].sum()), "fs_branch": int((fw["branch"] == 1).sum()),
               "std_branch": int((fw["branch"] == 2).sum()), "domains": int(len(dm)), "reported": int(dm["reported"].sum()),
               "envelope_nt": int((np.abs(dm["jenv"].astype(np.int64) - dm["ienv"]) + 1).sum()), "clustered_regions": int(nskip),
               "shifted_codons_found": int(dm["n_shifted_codons"].sum())}
    alone = None if os.environ.get("BATH_BENCH_NO_ALONE") == "1" else fs_envelopes_alone(ba, ctx, om5, flat, offsets, fw, dm, hmm.M)
    keys = ("window", "strand", "ienv", "jenv", "iali", "jali", "ihmm", "jhmm", "n_shifted_codons")
    strict = {tuple(int(r[k]) for k in keys) for r in dm}
    strict_bits = {tuple(int(r[k]) for k in keys) + (fbits(r["envsc"]),) for r in dm}
    conc = None if args.no_concurrent else fs_concurrent_leg(ba, hmm, flat, offsets, strict_bits, dt * 1e3, main_ctx=ctx if os.environ.get("BATH_BENCH_NO_TRIM") != "1" else None)
    # the same pass in the fast mode (sums along the model by wavefront scans: scores within O(1e-3) nats, outside the 1e-4 contract near zero)
    ctx.set_fs_strict(False)
    pipe.run_frameshift_domains(om3, om5, dna, arrays=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        f_stats, f_fw, f_dm, _ = pipe.run_frameshift_domains(om3, om5, dna, arrays=True)
    dtf = (time.perf_counter() - t0) / steps
    fast = {tuple(int(r[k]) for k in keys) for r in f_dm}
    n_fast = int(len(f_dm))
    ctx.set_fs_strict(True)
    return {
        "workload": "Caudal_act.bhmm (M=%d) --fs vs %d x %d nt windows, 1%% planted domains with indels (P(+-1 nt) = 0.01, P(+-2) = 0.005 per codon) "
                    "and in-frame stops (0.002), both strands: cascade (F4) -> DNA windows -> 3-codon parsers -> regions -> 5-codon "
                    "Forward/Backward/decoding/optimal accuracy/null2 -> traceback -> hits" % (hmm.M, args.fs_windows, args.length),
        "ms_per_pass": dt * 1e3, "residues_per_s": stats.nres / dt, "steps": steps,
        "parity_check": parity, "concurrent_blocks": conc, "cpu_baseline": (cpu or {}).get("baseline"),
        **summary,
        "kernels": kt,
        "roofline": {"bound": "hbm", "binding_resource": "memory requests issued (a lane writes its own row: 64 pieces per store instruction)", "kernels": env, "cells": env_cells, "bytes_per_cell": env_bytes / env_cells if env_cells else None,
                     "ms": env_ms, "achieved": env_bytes / (env_ms * 1e-3) / 1e9 if env_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": env_bytes / (env_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if env_ms > 0 else None,
                     "alone": alone,
                     "note": "5-codon envelope kernels: algorithmic matrix bytes (Forward writes 32, Backward writes 12 + 4 for the B terms, the fused decoding + "
                             "optimal-accuracy pass reads 44 and writes 12 per cell -- the posterior matrix is not written in pipeline mode, the traceback forms the posteriors it reads) / sum of their device times.  Forward and Backward are row-per-lane "
                             "wavefronts (bath_fs_wavefront.hip): a lane writes its own row, so their stores are 12-32 B pieces of 64 different lines per "
                             "instruction -- bound by memory requests issued, not by bytes (DESIGN.md 4.6)"},
        "mode": "strict (the library's default): every sum along the model in the reference's serial order; scores, special-state rows and matrices "
                "bit-identical to generic_fwdback_frameshift.c (tests/test_frameshift_gpu.py, tests/test_fs_strict_gpu.py)",
        "fast": {"what": "bath_hip_set_fs_strict(0): the same table log-sums associated by wavefront scans (scores within O(1e-3) nats of the reference: "
                         "outside the 1e-4 contract for scores near zero; profiles/r03_fs_fast_errors.json)",
                 "ms_per_pass": dtf * 1e3, "residues_per_s": f_stats.nres / dtf, "domains": n_fast, "domains_identical_to_strict_mode": len(fast & strict)},
    }


if __name__ == "__main__":
    main()
