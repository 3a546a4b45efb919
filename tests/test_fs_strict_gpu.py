"""The --fs pipeline in its default (strict) mode against the oracle with NO allowance on the frameshift branch.

In strict mode every sum along the model runs in the reference's serial order (bath_fs_chain.hip, bath_fs_wavefront.hip), so the
3-codon parsers' scores and special-state rows, the regions' Forward matrices and the envelopes' Forward/Backward matrices are
the oracle's bit for bit.  What follows from bit-identical inputs must be identical too:
  * every DNA window: coordinates, the frameshift Forward score (bitwise), its P-values, the branch it takes;
  * every domain of the frameshift branch: envelope, alignment and model coordinates and the shifted-codon count EXACTLY -- no
    "an envelope end may move one step", and the clustered regions sample for sample (the stochastic tracebacks walk identical
    matrices with the same random-number stream) -- and the envelope score bitwise.
The domains of the standard branch (windows the decision sends to p7_Forward/p7_Backward in fp32 odds-ratio arithmetic,
p7_pipeline.c:1479-1510) keep the tolerances of tests/test_fs_pipeline_gpu.py: that arithmetic is not table log-sums."""
import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol
import test_fs_pipeline_gpu as P

pytestmark = pytest.mark.gpu


def bits(x):
    return int(np.float32(x).view(np.uint32))


def run(ctx, path, wins):
    ctx.set_fs_strict(True)
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(ctx, wins))
    model = ol.Model(path, 0)
    pli, ofw, per_w, odm, per_d, oskip = model.run_pipeline_fsdom(wins)
    return model, stats, fw, dm, nskip, pli, ofw, per_w, odm, per_d, oskip


def check_exact(model, stats, fw, dm, nskip, pli, ofw, per_w, odm, per_d, oskip):
    # ---- DNA windows
    want = sorted(((w, o) for w, (a, b) in enumerate(per_w) for o in ofw[a:b]), key=lambda t: (t[0], t[1].strand, t[1].n))
    got = sorted(fw, key=lambda g: (g.window, g.strand, g.n))
    assert len(got) == len(want)
    for g, (w, o) in zip(got, want):
        assert (g.window, g.strand, g.n, g.length, g.orf_cnt, g.k_min, g.k_max) == (w, o.strand, o.n, o.length, o.orf_cnt, o.k_min, o.k_max)
        assert bits(g.fwdsc) == bits(o.fwdsc), (w, g.fwdsc, o.fwdsc)                      # p7_ForwardParser_Frameshift_3Codons, bit for bit
        assert abs(g.filtersc - o.filtersc) <= 1e-4 * max(1.0, abs(o.filtersc))             # the bias filter is an fp32 Forward of a 2-state HMM
        borderline = abs(o.P_null - o.P_tot) <= 1e-3 * max(o.P_null, o.P_tot)               # P_tot sums fp32 odds-ratio Forward scores (1e-4)
        if not borderline:
            assert g.branch == o.branch, (w, g.P_fs, o.P_fs)
    assert nskip == oskip
    # ---- domains: the frameshift branch exactly, the standard branch at the tolerances of its fp32 arithmetic
    key = lambda w, d: (w, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm, d.n_shifted_codons)
    g_fs = [d for d in dm if fw[d.fs_window].branch == 1]
    g_std = [d for d in dm if fw[d.fs_window].branch != 1]
    o_all = [(w, o) for w, (a, b) in enumerate(per_d) for o in odm[a:b]]
    assert len(dm) == len(o_all)
    omap = {}
    for w, o in o_all:
        omap.setdefault(key(w, o), []).append(o)
    used = set()
    for d in g_fs:
        lst = omap.get(key(d.window, d))
        assert lst, ("a frameshift-branch domain the oracle does not have", key(d.window, d))
        o = lst.pop()
        used.add(id(o))
        assert bits(d.envsc) == bits(o.envsc), (d.window, d.envsc, o.envsc)                 # p7_Forward_Frameshift of the envelope, bit for bit
        assert abs(d.oasc - o.oasc) <= 5e-4 + 1e-4 * abs(o.oasc)                             # posteriors differ by expf's last bit
        assert abs(d.bitscore - o.bitscore) <= 2e-3 and abs(d.domcorrection - o.domcorrection) <= 2e-3 + 2e-3 * abs(o.domcorrection)
        assert abs(d.lnP - o.lnP) <= 2e-3 and abs(d.pre_score - o.pre_score) <= 2e-3
        assert d.reported == o.reported, (d.window, d.lnP, o.lnP)                            # the early E-value test: pli->nres at this window and strand
    # what is left on both sides are the standard branch's domains
    odm_rest, per_rest = [], []
    for w, (a, b) in enumerate(per_d):
        lo = len(odm_rest)
        odm_rest += [o for o in odm[a:b] if id(o) not in used]
        per_rest.append((lo, len(odm_rest)))
    assert len(odm_rest) == len(g_std)
    if g_std:
        P.compare_domains(model, g_std, odm_rest, per_rest, nskip)
    return len(g_fs), len(g_std), len(dm)


@pytest.mark.parametrize("name", ["Caudal_act.bhmm", "PTH2.bhmm", "2OG-FeII_Oxy_3.bhmm"])
def test_strict_pipeline_is_exact_on_planted_frameshifted_genes(gpu_ctx, name):
    rng = np.random.default_rng(41)
    path = ol.GOLDEN + "/" + name
    wins = P.frameshifted_windows(rng, ol.Model(path, 0), n=24)
    out = run(gpu_ctx, path, wins)
    n_fs, n_std, n_all = check_exact(*out)
    assert n_fs >= 3 and n_std >= 1                        # both branches produced domains


def test_strict_pipeline_is_exact_on_clustered_regions(gpu_ctx):
    """Windows carrying two copies of a gene a short spacer apart: multi-domain regions, resolved by 200 stochastic tracebacks
    through the region's multihit Forward matrix (p7_domaindef.c:396-455).  With identical matrices both sides draw the same
    samples: the clusters' envelopes are identical, not merely near."""
    rng = np.random.default_rng(7)
    path = ol.GOLDEN + "/PTH2.bhmm"
    model = ol.Model(path, 0)
    genes = common.emit_from_model(rng, model, 12, flank=3, sharpen=2.0)
    wins = []
    for a, b in zip(genes[::2], genes[1::2]):
        nt = [list(common.revtranslate(rng, g, model.basic)) for g in (a, b)]
        for seq in nt:
            p = int(rng.integers(10, len(seq) - 10))
            del seq[p]                                         # one frameshift per copy
        wins.append(np.array(nt[0] + list(rng.integers(0, 4, size=int(rng.integers(20, 60)))) + nt[1], dtype=np.uint8))
    out = run(gpu_ctx, path, wins)
    assert out[4] >= 1, "no clustered region in this input"
    n_fs, n_std, n_all = check_exact(*out)
    assert n_fs >= 4


@pytest.mark.parametrize("M", [300, 513, 1024, 1200])
def test_strict_pipeline_is_exact_with_a_long_model(gpu_ctx, tmp_path, M):
    """BASELINE configs[4]'s model size (16 nodes per lane in the chain kernels, the wavefront's ring in global memory) and one
    beyond it (1200 nodes: the 20-nodes-per-lane instantiation -- configs[4] is not the edge of what the kernels take); 300 and
    513 nodes: the multi-wave decoding + optimal-accuracy kernel with 3 and 5 waves per envelope (8 at 1024, 7 x 3 nodes at 1200)."""
    path = common.write_synthetic_bhmm(str(tmp_path / ("s%d.bhmm" % M)), M, seed=M)
    rng = np.random.default_rng(12)
    wins = P.frameshifted_windows(rng, ol.Model(path, 0), n=8, L_flank=60)[:10] + common.random_dna(rng, 4, 1200)
    out = run(gpu_ctx, path, wins)
    n_fs, n_std, n_all = check_exact(*out)
    assert n_fs >= 2


def test_fast_mode_errors_on_scores_that_reach_output(gpu_ctx):
    """The fast mode (bath_hip_set_fs_strict(0): wavefront scans) is not the parity mode: its sums differ from the reference's by
    O(1e-3) nats, which is more than 1e-4 RELATIVE for scores near zero (tests/test_frameshift_gpu.py records up to 2.9e-4 over
    all scores above one nat).  The scores that can reach output are larger: the frameshift Forward score of every DNA window
    that passes F3 and the envelope score of every reported frameshift-branch domain must be within the north star's 1e-4
    relative of the oracle's; the worst cases go to gpurun_out/fs_fast_errors.json (committed as profiles/r03_fs_fast_errors.json)."""
    import json, os
    worst = {"window_fwdsc": {"n": 0, "max_rel": 0.0, "max_abs_nats": 0.0, "min_score_nats": 1e30},
             "reported_envsc": {"n": 0, "max_rel": 0.0, "max_abs_nats": 0.0, "min_score_nats": 1e30}}
    gpu_ctx.set_fs_strict(False)
    try:
        for name in ("Caudal_act.bhmm", "PTH2.bhmm", "2OG-FeII_Oxy_3.bhmm"):
            rng = np.random.default_rng(43)
            path = ol.GOLDEN + "/" + name
            model = ol.Model(path, 0)
            wins = P.frameshifted_windows(rng, model, n=40)
            hmm = ba.HMM(path, 0)
            om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
            om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
            om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
            pipe = ba.Pipeline(gpu_ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
            stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(gpu_ctx, wins))
            pli, ofw, per_w, odm, per_d, oskip = model.run_pipeline_fsdom(wins)
            omap = {(w, o.strand, o.n): o for w, (a, b) in enumerate(per_w) for o in ofw[a:b]}
            for g in fw:
                o = omap.get((g.window, g.strand, g.n))
                if o is None or o.P_fs > 1e-5:
                    continue
                e = worst["window_fwdsc"]
                e["n"] += 1; e["max_abs_nats"] = max(e["max_abs_nats"], abs(g.fwdsc - o.fwdsc)); e["min_score_nats"] = min(e["min_score_nats"], abs(o.fwdsc))
                e["max_rel"] = max(e["max_rel"], abs(g.fwdsc - o.fwdsc) / abs(o.fwdsc))
            key = lambda w, d: (w, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm)
            dmap = {key(w, o): o for w, (a, b) in enumerate(per_d) for o in odm[a:b]}
            for d in dm:
                o = dmap.get(key(d.window, d))
                if o is None or not o.reported or fw[d.fs_window].branch != 1:
                    continue
                e = worst["reported_envsc"]
                e["n"] += 1; e["max_abs_nats"] = max(e["max_abs_nats"], abs(d.envsc - o.envsc)); e["min_score_nats"] = min(e["min_score_nats"], abs(o.envsc))
                e["max_rel"] = max(e["max_rel"], abs(d.envsc - o.envsc) / abs(o.envsc))
    finally:
        gpu_ctx.set_fs_strict(True)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump({"mode": "fast (bath_hip_set_fs_strict(0))", "contract": "1e-4 relative (BASELINE.json north_star)", **worst}, open(os.path.join(out, "fs_fast_errors.json"), "w"), indent=1)
    print("fast-mode errors", worst)
    assert worst["window_fwdsc"]["n"] >= 30 and worst["reported_envsc"]["n"] >= 15
    assert worst["reported_envsc"]["max_rel"] <= 1e-4 and worst["window_fwdsc"]["max_rel"] <= 1e-4


BESIDE_SCRIPT = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np
import bath_amd as ba, common, oracle_lib as ol
import test_fs_strict_gpu as S
rng = np.random.default_rng(7)
path = ol.GOLDEN + "/PTH2.bhmm"
model = ol.Model(path, 0)
genes = common.emit_from_model(rng, model, 12, flank=3, sharpen=2.0)
wins = []
for a, b in zip(genes[::2], genes[1::2]):
    nt = [list(common.revtranslate(rng, g, model.basic)) for g in (a, b)]
    for seq in nt:
        del seq[int(rng.integers(10, len(seq) - 10))]
    wins.append(np.array(nt[0] + list(rng.integers(0, 4, size=int(rng.integers(20, 60)))) + nt[1], dtype=np.uint8))
wins += S.P.frameshifted_windows(rng, model, n=12)
ctx = ba.Context(0)                      # the process's ONLY context, as in a production host: host_contexts() == 1
out = S.run(ctx, path, wins)
assert out[4] >= 1, "no clustered region"
n_fs, n_std, n_all = S.check_exact(*out)
assert n_fs >= 4
out2 = S.run(ctx, path, wins)            # and again on the warm context (aux3 exists by now)
assert S.check_exact(*out2) == (n_fs, n_std, n_all)
print("beside ok", out[4], n_fs, n_std)
"""


@pytest.mark.parametrize("beside", ["1", "0", None])
def test_clusters_envelopes_beside_the_first_batch_fresh_process(beside):
    """The clusters' envelope batch on a context of its own, driven from the ensembles' thread beside the tail of the single-domain
    batch: ON by default only when the host holds ONE context -- which a pytest process with module-scoped contexts never does.
    Fresh processes with one context: forced on, forced off, and the default (on), each exact on a block with clustered regions."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("BATH_HIP_FS_CLUSTERS_BESIDE", None)
    if beside is not None:
        env["BATH_HIP_FS_CLUSTERS_BESIDE"] = beside
    r = subprocess.run([sys.executable, "-c", BESIDE_SCRIPT.format(root=root)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "beside ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def _rerun(env_extra, select):
    """Runs a selection of this module's tests in a FRESH process with extra environment (the switches are read once per process)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(here, "test_fs_strict_gpu.py"), os.path.join(here, "test_fs_pipeline_gpu.py"), "-k", select],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=os.path.dirname(here))
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])


def test_host_window_path_agrees_with_the_oracle_too():
    """The DNA windows are built and the branch decided on the device by default (bath_fs_windows.hip); the host path of rounds 1-5
    (p7_pli_BuildDNAWindows restated in C++, bath_pipeline.hip) stays as the fallback for blocks the device path does not take and as
    the A/B twin: BATH_HIP_FS_WINDOWS_HOST=1 runs the strict pipeline tests and the window-level tests through it."""
    _rerun({"BATH_HIP_FS_WINDOWS_HOST": "1"}, "strict_pipeline_is_exact or planted_frameshifted or cascade_lanes or recorded_fs")


def test_speculative_backward_rows_are_the_parsers_rows():
    """fs3_backward_spec (off by default: profiles/r06_fs_spec_probe.txt): the Backward parser of the K longest DNA windows beside the
    Forward parser of all of them, its rows copied into place for the domain stage, which runs the parser for the other windows only.
    Forced for a handful of windows (BATH_HIP_FS_SPEC_FORCE=1, K = 5): every strict pipeline test must still hold bit for bit."""
    _rerun({"BATH_HIP_FS_SPEC_FORCE": "1", "BATH_HIP_FS_SPEC_K": "5"}, "strict_pipeline_is_exact or planted_frameshifted or cascade_lanes or recorded_fs")


def test_blocks_the_device_window_path_refuses_fall_back_to_the_host_path():
    """bath_fs_windows.hip hands a block back (BATH_ENORESULT) when a (sequence, strand) group has more ORFs than it serialises in one
    thread; the pipeline then fetches the survivors and runs the host path.  BATH_HIP_FSW_MAX_GROUP=1 makes every input with two
    surviving ORFs on one strand such a block: the strict pipeline tests must hold through the fallback."""
    _rerun({"BATH_HIP_FSW_MAX_GROUP": "1"}, "strict_pipeline_is_exact or planted_frameshifted or cascade_lanes or recorded_fs")


def test_envelope_b_sums_by_a_wave_per_envelope_too():
    """fs5_bwd_x_kernel adds up B(i) of an envelope's rows with the waves of a block sharing ONE envelope when a launch has few envelopes
    (every test input) and with a wave per envelope otherwise (the bench block's thousands).  BATH_HIP_FS_BWDX_TEAM=0 runs the strict
    pipeline tests through the wave-per-envelope form."""
    _rerun({"BATH_HIP_FS_BWDX_TEAM": "0"}, "strict_pipeline_is_exact or planted_frameshifted or recorded_fs")
