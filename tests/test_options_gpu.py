"""The pipeline's option state beyond F1-F4 / --nobias (p7_pipeline_Create_BATH, p7_pipeline.c:94-234; bathsearch.c:718-719,
:831-833), one test per switch, GPU path against the oracle run with the same switch, on planted inputs, in both branches
of the pipeline where the switch applies:

  --nonull2   do_null2 = 0   a hit's bias correction is 0                                  (p7_pipeline.c:1063, :1230)
  --fsonly    std_pipe = 0   P_tot = 1 in the branch decision, no standard branch          (:1457, :1480)
  --strand    strands        only that strand is translated, searched and counted          (bathsearch.c:1069, :1082)
  -m / -M     initiator      ORFs start at AUG / at the codon table's start codons, read M (bathsearch.c:718-719)
  -T, --incT  inc_by_E, T    the early reporting test of the domain stage goes by bit score (:1080, :1247; p7_domaindef.c:1034)
  --seed      seed           the generator of the stochastic-trace ensembles               (:98, :140-143; p7_domaindef.c:781, :904)

The comparisons are the ones of tests/test_hits_gpu.py (standard pipeline) and tests/test_fs_strict_gpu.py (--fs: windows,
frameshift-branch domains exact), so every default-option test keeps guarding the defaults."""
import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol
import test_fs_pipeline_gpu as P
import test_fs_strict_gpu as S
import test_hits_gpu as H
import test_translate_gpu as T

pytestmark = pytest.mark.gpu

CAUDAL = ol.GOLDEN + "/Caudal_act.bhmm"
PTH2 = ol.GOLDEN + "/PTH2.bhmm"


def std_inputs(model, seed=123, n=24, tandem=True):
    """Planted genes on both strands (some truncated, some in tandem: clustered regions) and background."""
    rng = np.random.default_rng(seed)
    genes = common.emit_from_model(rng, model, n // 2, flank=5) + common.emit_from_model(rng, model, n - n // 2, flank=5, sharpen=2.0)
    wins = []
    for i, aa in enumerate(genes):
        nt = np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8)
        w = np.concatenate([rng.integers(0, 4, size=int(rng.integers(0, 300))).astype(np.uint8), nt,
                            rng.integers(0, 4, size=int(rng.integers(0, 300))).astype(np.uint8)])
        wins.append((3 - w[::-1]).astype(np.uint8) if i % 2 else w)
    if tandem:
        for i in range(0, 8, 2):
            nt = np.array(common.revtranslate(rng, list(genes[i]) + list(genes[i + 1]), model.basic), dtype=np.uint8)
            w = np.concatenate([rng.integers(0, 4, size=33).astype(np.uint8), nt, rng.integers(0, 4, size=60).astype(np.uint8)])
            wins.append((3 - w[::-1]).astype(np.uint8) if i % 4 else w)
    return wins + common.random_dna(rng, 20, 1000)


def run_std(ctx, path, wins, opts, E=10.0):
    model = ol.Model(path, 0)
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct, **opts)
    stats, dm, nskip = pipe.run_hits(ba.SeqBlock(ctx, wins), E_report=E)
    pli, odm, per_d, onskip = model.run_pipeline_hits(wins, E=E, opts=opts)
    for f in ("nres", "n_orfs", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd", "pos_past_msv", "pos_past_bias", "pos_past_vit", "pos_past_fwd"):
        assert getattr(stats, f) == getattr(pli, f), (f, getattr(stats, f), getattr(pli, f))
    n = H.compare_hits(dm, odm, per_d, nskip, onskip)
    return model, stats, dm, nskip, pli, odm, n


def run_fs(ctx, path, wins, opts, E=10.0):
    ctx.set_fs_strict(True)
    model = ol.Model(path, 0)
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct, **opts)
    stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(ctx, wins), E_report=E)
    pli, ofw, per_w, odm, per_d, oskip = model.run_pipeline_fsdom(wins, E=E, opts=opts)
    for f in ("nres", "n_orfs", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd", "pos_past_msv", "pos_past_bias", "pos_past_vit", "pos_past_fwd"):
        assert getattr(stats, f) == getattr(pli, f), (f, getattr(stats, f), getattr(pli, f))
    counts = S.check_exact(model, stats, fw, dm, nskip, pli, ofw, per_w, odm, per_d, oskip)
    return model, stats, fw, dm, nskip, pli, ofw, odm, counts


# ---------------------------------------------------------------------------------------------- --nonull2
def test_nonull2_standard_pipeline(gpu_ctx):
    wins = std_inputs(ol.Model(CAUDAL, 0))
    _, _, dm, _, _, odm, n = run_std(gpu_ctx, CAUDAL, wins, {"do_null2": 0})
    assert n >= 10
    assert all(d.dombias == 0.0 for d in dm) and all(o.dombias == 0.0 for o in odm)
    _, _, dm1, _, _, _, _ = run_std(gpu_ctx, CAUDAL, wins, {})
    assert any(d.dombias > 0.0 for d in dm1)                       # the default does correct: the switch is what removed it
    k = lambda d: (d.window, d.ienv, d.jenv)
    b0, b1 = {k(d): d.bitscore for d in dm}, {k(d): d.bitscore for d in dm1}
    assert any(b0[x] > b1[x] for x in b0 if x in b1)               # ... and without the correction a hit scores higher


def test_nonull2_frameshift_pipeline(gpu_ctx):
    rng = np.random.default_rng(41)
    wins = P.frameshifted_windows(rng, ol.Model(CAUDAL, 0), n=24)
    _, _, fw, dm, _, _, _, odm, (n_fs, n_std, _) = run_fs(gpu_ctx, CAUDAL, wins, {"do_null2": 0})
    assert n_fs >= 3 and n_std >= 1
    assert all(d.dombias == 0.0 for d in dm) and all(o.dombias == 0.0 for o in odm)


# ---------------------------------------------------------------------------------------------- --fsonly
def test_fsonly(gpu_ctx):
    rng = np.random.default_rng(41)
    wins = P.frameshifted_windows(rng, ol.Model(CAUDAL, 0), n=24)
    _, stats, fw, dm, _, pli, ofw, _, (n_fs, n_std, _) = run_fs(gpu_ctx, CAUDAL, wins, {"std_pipe": 0})
    assert n_std == 0 and n_fs >= 3
    assert all(w.P_tot == 1.0 for w in fw) and all(w.P_tot == 1.0 for w in ofw)
    assert {w.branch for w in fw} <= {0, 1}
    # with the standard branch gone, windows that took it by default take the frameshift branch or none
    _, _, fw1, dm1, _, _, _, _, (n_fs1, n_std1, _) = run_fs(gpu_ctx, CAUDAL, wins, {})
    assert n_std1 >= 1 and sum(w.branch == 1 for w in fw) >= sum(w.branch == 1 for w in fw1)


# ---------------------------------------------------------------------------------------------- --strand
@pytest.mark.parametrize("strands", [ba.STRAND_TOPONLY, ba.STRAND_BOTTOMONLY])
def test_one_strand_standard_pipeline(gpu_ctx, strands):
    wins = std_inputs(ol.Model(CAUDAL, 0))
    _, stats, dm, _, _, _, n = run_std(gpu_ctx, CAUDAL, wins, {"strands": strands})
    _, stats2, dm2, _, _, _, n2 = run_std(gpu_ctx, CAUDAL, wins, {})
    assert stats.nres * 2 == stats2.nres                            # a window's W once per strand searched
    want = 0 if strands == ba.STRAND_TOPONLY else 1
    assert n >= 4 and all(d.strand == want for d in dm)
    assert any(d.strand != want for d in dm2)
    k = lambda d: (d.window, d.ienv, d.jenv, d.iali, d.jali)
    assert {k(d) for d in dm} == {k(d) for d in dm2 if d.strand == want}      # coordinates do not depend on the other strand


@pytest.mark.parametrize("strands", [ba.STRAND_TOPONLY, ba.STRAND_BOTTOMONLY])
def test_one_strand_frameshift_pipeline(gpu_ctx, strands):
    rng = np.random.default_rng(41)
    wins = P.frameshifted_windows(rng, ol.Model(PTH2, 0), n=24)
    _, stats, fw, dm, _, _, _, _, (n_fs, n_std, n_all) = run_fs(gpu_ctx, PTH2, wins, {"strands": strands})
    want = 0 if strands == ba.STRAND_TOPONLY else 1
    assert n_all >= 4 and all(w.strand == want for w in fw) and all(d.strand == want for d in dm)


def test_one_strand_translation(gpu_ctx):
    rng = np.random.default_rng(9)
    windows = [T.rand_dna(rng, n) for n in (14, 15, 400, 1000, 1153, 5000)] + [T.rand_dna(rng, 3000, stop_poor=True)]
    dna = ba.SeqBlock(gpu_ctx, windows)
    both = ba.translate_orfs(gpu_ctx, dna)
    for strands, want in ((ba.STRAND_TOPONLY, 0), (ba.STRAND_BOTTOMONLY, 1)):
        got = ba.translate_orfs(gpu_ctx, dna, strands=strands)
        ref = [o for o in both if o[1] == want]
        assert len(got) == len(ref) and len(got) > 10
        for g, o in zip(got, ref):
            assert tuple(g[:5]) == tuple(o[:5]) and np.array_equal(g[5], o[5])


# ---------------------------------------------------------------------------------------------- -m / -M
def oracle_orfs_init(windows, ct, minlen, initiator):
    import ctypes as C
    L_ = ol.lib()
    basic = np.zeros(64, np.uint8); is_init = np.zeros(64, np.uint8)
    assert L_.bo_gencode_basic(ct, ol.u8(basic)) == 0 and L_.bo_gencode_initiators(ct, initiator, ol.u8(is_init)) == 0
    out = []
    blk = ol.OrfBlock(); L_.bo_orfblock_init(C.byref(blk))
    for w, codes in enumerate(windows):
        n = len(codes)
        if n < 15:
            continue
        d = ol.dsq_from(codes)
        rc = np.zeros(n + 2, np.uint8)
        L_.bo_revcomp(ol.u8(d), n, ol.u8(rc))
        for strand, dsq in ((0, d), (1, rc)):
            L_.bo_orfblock_reuse(C.byref(blk))
            L_.bo_translate_orfs_init(ol.u8(dsq), n, ol.u8(basic), ol.u8(is_init) if initiator else None, 1 if initiator else 0, minlen, C.byref(blk))
            if blk.count == 0:
                continue
            aa = np.ctypeslib.as_array(blk.aa, shape=(int(blk.aa_n),))
            for i in range(blk.count):
                o = blk.orf[i]
                out.append((w, strand, o.frame, o.start, o.end, aa[o.off + 1:o.off + 1 + o.n].copy()))
    L_.bo_orfblock_free(C.byref(blk))
    out.sort(key=lambda r: r[:4])
    return out


@pytest.mark.parametrize("initiator,ct,minlen", [(ba.INIT_AUG, 1, 20), (ba.INIT_TABLE, 1, 20), (ba.INIT_TABLE, 4, 20), (ba.INIT_TABLE, 11, 5),
                                                  (ba.INIT_AUG, 1, 1), (ba.INIT_TABLE, 2, 30)])
def test_initiation_codons_translation(gpu_ctx, initiator, ct, minlen):
    """ORF by ORF: coordinates and residues (the initiation codon reads M), windows around the tile sizes, ORFs crossing tiles
    (stop-poor DNA), degenerate nucleotides (a degenerate codon initiates only if all its expansions do)."""
    rng = np.random.default_rng(100 + initiator + ct)
    windows = [T.rand_dna(rng, n) for n in (14, 15, 16, 17, 383, 384, 385, 400, 1000, 1151, 1152, 1153, 5000)]
    windows += [T.rand_dna(rng, 3000, stop_poor=True), T.rand_dna(rng, 20000, stop_poor=True), T.rand_dna(rng, 2000, p_degen=0.02)]
    dna = ba.SeqBlock(gpu_ctx, windows)
    got = ba.translate_orfs(gpu_ctx, dna, ct, minlen, initiator=initiator)
    want = oracle_orfs_init(windows, ct, minlen, initiator)
    assert len(got) == len(want) and len(got) > 20, (len(got), len(want))
    for g, o in zip(got, want):
        assert tuple(g[:5]) == tuple(o[:5]), (g[:5], o[:5])
        assert np.array_equal(g[5], o[5]), (g[:5],)
        assert g[5][0] == 10                                        # M
    any_orfs = ba.translate_orfs(gpu_ctx, dna, ct, minlen)
    assert len(any_orfs) > len(got)                                 # the requirement does remove and shorten ORFs


@pytest.mark.parametrize("initiator", [ba.INIT_AUG, ba.INIT_TABLE])
def test_initiation_codons_pipelines(gpu_ctx, initiator):
    """Genes planted behind an ATG (so that -m keeps them) through both pipelines."""
    model = ol.Model(CAUDAL, 0)
    rng = np.random.default_rng(5)
    wins = []
    for i, aa in enumerate(common.emit_from_model(rng, model, 16, flank=5, sharpen=2.0)):
        nt = list(common.revtranslate(rng, [10] + list(aa), model.basic))          # M first: its only codon is ATG
        if i % 3 == 0:
            del nt[len(nt) // 2]                                                    # a frameshift in every third gene
        w = np.concatenate([rng.integers(0, 4, size=int(rng.integers(3, 200))).astype(np.uint8), np.array(nt, dtype=np.uint8),
                            rng.integers(0, 4, size=int(rng.integers(0, 200))).astype(np.uint8)])
        wins.append((3 - w[::-1]).astype(np.uint8) if i % 2 else w)
    wins += common.random_dna(rng, 20, 1000)
    _, stats, dm, _, pli, _, n = run_std(gpu_ctx, CAUDAL, wins, {"initiator": initiator})
    _, stats_any, _, _, _, _, _ = run_std(gpu_ctx, CAUDAL, wins, {})
    assert n >= 5 and stats.n_orfs < stats_any.n_orfs
    _, _, fw, dm, _, _, _, _, (n_fs, n_std, n_all) = run_fs(gpu_ctx, CAUDAL, wins, {"initiator": initiator})
    assert n_all >= 5


# ---------------------------------------------------------------------------------------------- -T / --incT
def test_inclusion_by_score_standard_pipeline(gpu_ctx):
    """--incT clears inc_by_E: the early test flags a hit by its bit score against pli->T (0 unless -T is given) -- p7_pipeline.c:1247."""
    wins = std_inputs(ol.Model(CAUDAL, 0), tandem=False)
    _, _, dm0, _, _, _, _ = run_std(gpu_ctx, CAUDAL, wins, {"inc_by_E": 0, "T": 0.0})
    scores = sorted(d.bitscore for d in dm0)
    T_mid = float(scores[len(scores) // 2]) + 0.05
    _, _, dm, _, _, odm, n = run_std(gpu_ctx, CAUDAL, wins, {"inc_by_E": 0, "T": T_mid})
    assert n >= 6
    rep = [d for d in dm if d.reported]
    assert 0 < len(rep) < len(dm) and all(d.bitscore >= T_mid for d in rep) and all(d.bitscore < T_mid for d in dm if not d.reported)
    assert sorted(o.reported for o in odm) == sorted(d.reported for d in dm)
    # -T alone leaves inc_by_E set: the early test still goes by E (the reference's own quirk), whatever T says
    _, _, dmT, _, _, _, _ = run_std(gpu_ctx, CAUDAL, wins, {"inc_by_E": 1, "T": 1000.0})
    assert all(d.reported for d in dmT)


def test_inclusion_by_score_frameshift_pipeline(gpu_ctx):
    """... and in the frameshift branch, where inc_by_E = 0 also turns off the early E-value drop of an envelope (p7_domaindef.c:1034)."""
    rng = np.random.default_rng(41)
    wins = P.frameshifted_windows(rng, ol.Model(CAUDAL, 0), n=24)
    _, _, fw, dm0, _, _, _, _, _ = run_fs(gpu_ctx, CAUDAL, wins, {"inc_by_E": 0, "T": 0.0}, E=1e-30)
    fsd = [d for d in dm0 if fw[d.fs_window].branch == 1]
    assert fsd                                                      # E = 1e-30 would have dropped every envelope before rescoring
    T_mid = float(np.median([d.bitscore for d in fsd]))
    _, _, fw, dm, _, _, _, odm, (n_fs, n_std, _) = run_fs(gpu_ctx, CAUDAL, wins, {"inc_by_E": 0, "T": T_mid}, E=1e-30)
    assert n_fs >= 3
    assert 0 < sum(d.reported for d in dm) < len(dm)
    _, _, fw, dmE, _, _, _, _, _ = run_fs(gpu_ctx, CAUDAL, wins, {}, E=1e-30)
    assert not [d for d in dmE if fw[d.fs_window].branch == 1]      # by E: dropped early


# ---------------------------------------------------------------------------------------------- --seed
def clustered_fs_inputs(path, seed=7):
    rng = np.random.default_rng(seed)
    model = ol.Model(path, 0)
    genes = common.emit_from_model(rng, model, 12, flank=3, sharpen=2.0)
    wins = []
    for a, b in zip(genes[::2], genes[1::2]):
        nt = [list(common.revtranslate(rng, g, model.basic)) for g in (a, b)]
        for seq in nt:
            del seq[int(rng.integers(10, len(seq) - 10))]
        wins.append(np.array(nt[0] + list(rng.integers(0, 4, size=int(rng.integers(20, 60)))) + nt[1], dtype=np.uint8))
    return wins


@pytest.mark.parametrize("seed", [7, 12345])
def test_seed_frameshift_clustered_regions(gpu_ctx, seed):
    """Both sides draw the region's 200 tracebacks from a generator started with --seed: identical envelopes (strict mode)."""
    wins = clustered_fs_inputs(PTH2)
    out = run_fs(gpu_ctx, PTH2, wins, {"seed": seed})
    assert out[4] >= 1 and out[8][0] >= 4


def test_seed_standard_clustered_regions(gpu_ctx):
    wins = std_inputs(ol.Model(PTH2, 0), seed=77)
    _, _, dm, nskip, _, _, n = run_std(gpu_ctx, PTH2, wins, {"seed": 2024})
    assert nskip >= 2 and n >= 8


def test_seed_zero_runs_without_reseeding(gpu_ctx):
    """--seed 0: an arbitrary one-time seed, not reproducible in the reference either (time of day): the run must work and
    resolve the same regions; which envelopes the clusters give is free."""
    wins = clustered_fs_inputs(PTH2)
    hmm = ba.HMM(PTH2, 0)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct)); om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    res = {}
    for seed in (0, 42):
        pipe = ba.Pipeline(gpu_ctx, om, fs_pipe=True, ncbi_table=hmm.ct, seed=seed)
        _, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(gpu_ctx, wins))
        res[seed] = (nskip, len(dm))
    assert res[0][0] == res[42][0] >= 1 and res[0][1] >= 4


def test_defaults_are_bathsearchs(gpu_ctx):
    p = ba.PipelineParams()
    ba.lib().bath_pipeline_params_default(p, 1)
    assert (p.do_null2, p.std_pipe, p.strands, p.initiator, p.inc_by_E, p.seed, p.T) == (1, 1, 0, 0, 1, 42, 0.0)
