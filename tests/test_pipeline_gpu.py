"""GPU parity of the filter cascade (bath_hip_pipeline_filters) against the oracle's restatement of
p7_Pipeline_BATH (oracle/pipeline.c), ORF by ORF, plus the reference's own recorded counters
(tutorial/*.out, tests/golden) reproduced end to end on the GPU."""
import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def make_windows(rng, model, n_random=150, n_planted=60, L=1000):
    basic = model.basic
    wins = common.random_dna(rng, n_random, L)
    planted = common.emit_from_model(rng, model, n_planted // 2, flank=10) + common.emit_from_model(rng, model, n_planted // 2, flank=10, sharpen=2.0)
    for i, aa in enumerate(planted):
        nt = common.revtranslate(rng, aa, basic)
        pre = rng.integers(0, 4, size=int(rng.integers(0, 200))).astype(np.uint8)
        post = rng.integers(0, 4, size=int(rng.integers(0, 200))).astype(np.uint8)
        w = np.concatenate([pre, nt, post])
        if i % 2:
            w = (3 - w[::-1]).astype(np.uint8)          # put it on the bottom strand
        wins.append(w)
    wins += common.random_dna(rng, 10, 300, degenerate_frac=0.02)
    wins += [rng.integers(0, 4, size=n).astype(np.uint8) for n in (0, 1, 14, 15, 16, 59, 60, 61, 62, 63, 3001)]
    return wins


def run_both(ctx, path, idx, wins, fs_pipe):
    model = ol.Model(path, idx)
    hmm = ba.HMM(path, idx)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(ctx, om, fs_pipe=fs_pipe, ncbi_table=hmm.ct)
    stats, res = pipe.run(ba.SeqBlock(ctx, wins))
    pli, ores, per_seq = model.run_pipeline(wins, fs_pipe=fs_pipe)
    return stats, res, pli, ores, per_seq, pipe


def compare(stats, res, pli, ores, per_seq):
    for name in ("nres", "n_orfs", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd",
                 "pos_past_msv", "pos_past_bias", "pos_past_vit", "pos_past_fwd", "cells_msv", "cells_vit", "cells_fwd"):
        assert getattr(stats, name) == getattr(pli, name), name
    want = {}
    for w, (a, b) in enumerate(per_seq):
        for r in ores[a:b]:
            if r.stage >= 1:
                want[(w, r.strand, r.frame, r.start)] = r
    assert len(res) == len(want)
    for g in res:
        o = want[(int(g["window"]), int(g["strand"]), int(g["frame"]), int(g["start"]))]
        assert (g["end"], g["n"], g["stage"], g["msv_status"]) == (o.end, o.n, o.stage, o.msv_status)
        assert np.float32(g["usc"]).view(np.uint32) == np.float32(o.usc).view(np.uint32)            # integer filter: bit exact
        assert np.float32(g["nullsc"]).view(np.uint32) == np.float32(o.nullsc).view(np.uint32)
        if o.stage >= 2 or (o.stage == 1):
            assert abs(g["filtersc"] - o.filtersc) <= 1e-4 + 1e-5 * abs(o.filtersc)
        if np.isfinite(o.vfsc) or np.isinf(g["vfsc"]):
            assert np.float32(g["vfsc"]).view(np.uint32) == np.float32(o.vfsc).view(np.uint32)      # integer filter: bit exact
            assert g["vit_status"] == o.vit_status
        if o.stage >= 3 and np.isfinite(o.fwdsc):
            assert abs(g["fwdsc"] - o.fwdsc) <= 2e-4 + 1e-4 * abs(o.fwdsc)


@pytest.mark.parametrize("name,fs", [("Caudal_act.bhmm", False), ("Caudal_act.bhmm", True), ("PTH2.bhmm", False), ("2OG-FeII_Oxy_3.bhmm", True)])
def test_cascade_matches_oracle(gpu_ctx, name, fs):
    rng = np.random.default_rng(7)
    path = ol.GOLDEN + "/" + name
    wins = make_windows(rng, ol.Model(path, 0))
    stats, res, pli, ores, per_seq, _ = run_both(gpu_ctx, path, 0, wins, fs)
    compare(stats, res, pli, ores, per_seq)
    assert stats.n_past_fwd > 0 and stats.n_past_msv > stats.n_past_fwd      # the set exercises every stage


def test_cascade_with_lane_kernels_forced(monkeypatch):
    """Blocks below 150 M nt (every test block) take the wave-per-ORF MSV / Viterbi kernels since round 4 (few candidates: bath_pipeline.hip,
    few_cands); the bench's blocks take the lane-per-ORF kernels with the length sort and the long-ORF split.  BATH_HIP_LANE_MIN_NT=0
    (read once per process) forces that path onto the test blocks: the oracle comparisons and the reference's recorded counters
    again, in a fresh process."""
    import os, subprocess, sys
    env = dict(os.environ, BATH_HIP_LANE_MIN_NT="0")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_cascade_matches_oracle or test_reference_recorded_counters or test_concurrent_lanes"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0 and " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-3000:]


GOLDEN_RUNS = [  # (model file, index, target fasta, counters printed by the reference: p7_pli_Statistics)
    ("PTH2.bhmm", 0, "target-PTH2.fa", (6000, 1503, 1503, 1401, 1287)),
    ("AMP_N.bhmm", 0, "target-AMP_N.fa", (822, 537, 537, 393, 237)),
    ("MET-ct4.bhmm", 0, "target-MET.fa", (71226, 2487, 2298, 666, 549)),
    ("MET-ct4.bhmm", 1, "target-MET.fa", (71226, 9168, 2775, 1062, 618)),
]


@pytest.mark.parametrize("hmmfile,idx,fasta,expect", GOLDEN_RUNS, ids=[g[0] + str(g[1]) for g in GOLDEN_RUNS])
def test_reference_recorded_counters(gpu_ctx, hmmfile, idx, fasta, expect):
    """tutorial/*.out 'Internal pipeline statistics summary' reproduced by the GPU pipeline."""
    hmm = ba.HMM(ol.GOLDEN + "/" + hmmfile, idx)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    seqs = [ba.digitize(s, ba.DNA_SYMS) for _, s in ol.read_fasta(ol.GOLDEN + "/" + fasta)]
    stats, _ = ba.Pipeline(gpu_ctx, om, fs_pipe=False, ncbi_table=hmm.ct).run(ba.SeqBlock(gpu_ctx, seqs))
    assert (stats.nres, stats.pos_past_msv, stats.pos_past_bias, stats.pos_past_vit, stats.pos_past_fwd) == expect


def test_nobias_and_thresholds(gpu_ctx):
    rng = np.random.default_rng(11)
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    model = ol.Model(path, 0)
    wins = make_windows(rng, model, 60, 30)
    hmm = ba.HMM(path)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(gpu_ctx, om, fs_pipe=False, do_biasfilter=0, F1=0.1, F2=0.01)
    stats, res = pipe.run(ba.SeqBlock(gpu_ctx, wins))
    import ctypes as C
    L = ol.lib()
    pli = ol.Pipeline(); L.bo_pipeline_init(C.byref(pli), 0)
    pli.do_biasfilter = 0; pli.F1 = 0.1; pli.F2 = 0.01
    resp = C.POINTER(ol.OrfResult)(); n = C.c_int(0); a = C.c_int(0)
    for w in wins:
        d = ol.dsq_from(w)
        L.bo_pipeline_window(C.byref(pli), model.om, model.sd, C.byref(model.bg), ol.u8(model.basic), ol.u8(d), len(w), C.byref(resp), C.byref(n), C.byref(a))
    for name in ("n_orfs", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd", "pos_past_fwd"):
        assert getattr(stats, name) == getattr(pli, name), name


def test_cascade_long_synthetic_model(gpu_ctx, tmp_path):
    """BASELINE config 5 shape: a 1024-node model does not fit one lane's registers; the SSV kernel splits
    the model over 4 adjacent lanes.  Same ORF-by-ORF parity bar."""
    rng = np.random.default_rng(5)
    path = common.write_synthetic_bhmm(str(tmp_path / "s1024.bhmm"), 1024, seed=1024)
    wins = make_windows(rng, ol.Model(path, 0), 40, 24, L=3500)
    stats, res, pli, ores, per_seq, _ = run_both(gpu_ctx, path, 0, wins, True)
    compare(stats, res, pli, ores, per_seq)
    assert stats.n_past_fwd > 0


@pytest.mark.parametrize("lanes", [2, 3])
def test_concurrent_lanes_give_identical_results(gpu_ctx, lanes, monkeypatch):
    """A block cut into parts that run the cascade concurrently on separate streams (bath_hip_pipeline_filters does this
    for large blocks) must return exactly what one pass over the whole block returns."""
    rng = np.random.default_rng(17)
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    wins = make_windows(rng, ol.Model(path, 0), n_random=300, n_planted=80)
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    dna = ba.SeqBlock(gpu_ctx, wins)
    pipe = ba.Pipeline(gpu_ctx, om, ncbi_table=hmm.ct)
    monkeypatch.setenv("BATH_HIP_LANES", "1")
    s1, r1 = pipe.run(dna)
    monkeypatch.setenv("BATH_HIP_LANES", str(lanes))
    s2, r2 = pipe.run(dna)
    for name, _ in ba.PipelineStats._fields_:
        assert getattr(s1, name) == getattr(s2, name), name
    assert len(r1) == len(r2)
    for f in r1.dtype.names:                              # field by field: the records carry padding bytes
        assert np.array_equal(r1[f], r2[f], equal_nan=True), f
    assert s1.n_past_fwd > 0


def test_window_buffer_overflow_is_retried(monkeypatch):
    """hit windows beyond the window buffer were dropped silently.  With a deliberately tiny buffer
    (BATH_HIP_TEST_WINCAP, read once per process: this test runs the pipeline in a child process) the cascade must notice,
    repeat the pass with room for every window, and give exactly the results of an ordinary run."""
    import subprocess, sys, os, json
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import bath_amd as ba, common, oracle_lib as ol
path = ol.GOLDEN + "/Caudal_act.bhmm"
model = ol.Model(path)
rng = np.random.default_rng(5)
wins = [common.revtranslate(rng, aa, model.basic) for aa in common.emit_from_model(rng, model, 60, flank=10, sharpen=2.0)]
ctx = ba.Context(0)
hmm = ba.HMM(path)
om = ba.OProfile(ctx, ba.Profile(hmm))
om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3))
pipe = ba.Pipeline(ctx, om, fs_pipe=True)
stats, res, fw = pipe.run_frameshift(om3, ba.SeqBlock(ctx, wins))
print(json.dumps({"fw": sorted((w.window, w.strand, w.n, w.length, w.orf_cnt, w.k_min, w.k_max, w.branch) for w in fw),
                  "stats": [stats.n_past_vit, stats.pos_past_vit, stats.n_past_fwd]}))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for cap in (None, "3"):
        env = dict(os.environ)
        env.pop("BATH_HIP_TEST_WINCAP", None)
        if cap:
            env["BATH_HIP_TEST_WINCAP"] = cap
        p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1] and len(outs[0]["fw"]) >= 30


def test_streamed_packed_blocks_equal_resident_blocks(gpu_ctx):
    """A block uploaded in 2 bits per nucleotide (+ exception list for degenerate codes), expanded on the device, must give the
    cascade exactly what the byte-per-nucleotide block gives: ragged lengths (not multiples of 4 or 16), degenerate nucleotides,
    a refill of the same block object with other content, page-locked and pageable sources."""
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    model = ol.Model(path)
    hmm = ba.HMM(path)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(gpu_ctx, om, fs_pipe=False)
    rng = np.random.default_rng(17)
    lens = [1000, 997, 15, 1, 0, 1003, 64, 4099, 70001, 131072 + 5] + [int(x) for x in rng.integers(200, 1400, size=40)]   # two longer than a wave's 64 k nt share

    def content(seed):
        r = np.random.default_rng(seed)
        seqs = [r.integers(0, 4, size=L).astype(np.uint8) for L in lens]
        for i, aa in enumerate(common.emit_from_model(r, model, 12, flank=5)):
            nt = common.revtranslate(r, aa, model.basic)
            j = 10 + i
            k = min(len(nt), len(seqs[j]))
            seqs[j][:k] = nt[:k]
        for j in (0, 5, 9, 20):                                   # degenerate nucleotides -> exception list
            idx = r.integers(0, len(seqs[j]), size=5)
            seqs[j][idx] = r.choice([5, 9, 15], size=5)
        return seqs
    offsets = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=offsets[1:])
    blk = ba.StreamedBlock(gpu_ctx, offsets)
    pin = ba.PinnedBuffer(int(sum((L + 3) // 4 for L in lens)))
    for round_, seed in enumerate((1, 2, 3)):
        seqs = content(seed)
        flat = np.concatenate(seqs)
        packed, es, ep, ec = ba.pack2(flat, offsets)
        assert len(es) >= 10
        if round_ < 2:
            pin.array[:len(packed)] = packed
            blk.upload(pin, es, ep, ec)
        else:
            blk.upload(packed, es, ep, ec)                        # pageable source
        blk.wait()
        st_s, res_s = pipe.run(blk)
        st_r, res_r = pipe.run(ba.SeqBlock(gpu_ctx, seqs))
        for f in ("nres", "n_orfs", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd", "pos_past_msv", "pos_past_bias", "pos_past_vit", "pos_past_fwd"):
            assert getattr(st_s, f) == getattr(st_r, f), f
        assert len(res_s) == len(res_r) and st_s.n_past_fwd >= 3
        for f in res_s.dtype.names:
            assert np.array_equal(res_s[f], res_r[f], equal_nan=True), f


def test_trim_releases_and_the_next_call_rebuilds(gpu_ctx):
    """bath_hip_trim: lanes, side contexts and side streams go away; the next pipeline calls create them again and give the same
    results (cascade counters, --fs domains)."""
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    ctx = ba.Context(0)
    hmm = ba.HMM(path)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    rng = np.random.default_rng(3)
    model = ol.Model(path)
    wins = common.random_dna(rng, 40, 900) + [common.revtranslate(rng, aa, model.basic) for aa in common.emit_from_model(rng, model, 12, flank=8)]
    blk = ba.SeqBlock(ctx, wins)
    pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    key = lambda dm: sorted((d.window, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm, np.float32(d.envsc).view(np.uint32)) for d in dm)
    s1, _, d1, n1 = pipe.run_frameshift_domains(om3, om5, blk)
    ctx.trim()
    s2, _, d2, n2 = pipe.run_frameshift_domains(om3, om5, blk)
    ctx.trim(); ctx.trim()
    c1, _ = ba.Pipeline(ctx, om, fs_pipe=False).run(blk)
    s3, _, d3, n3 = pipe.run_frameshift_domains(om3, om5, blk)
    assert key(d1) == key(d2) == key(d3) and len(d1) >= 4 and n1 == n2 == n3
    for f in ("nres", "n_orfs", "n_past_msv", "n_past_fwd", "pos_past_fwd"):
        assert getattr(s1, f) == getattr(s2, f) == getattr(s3, f)
    ctx.close()
