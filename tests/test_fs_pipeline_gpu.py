"""The frameshift stage of the pipeline on the GPU (bath_hip_pipeline_frameshift) against the oracle's restatement of
p7_pli_BuildDNAWindows + p7_pli_Frameshift (oracle/fs_pipeline.c), DNA window by DNA window, and against the one number
the reference recorded for this stage (tutorial/AMP_N-fs.out).

Integer outputs (window coordinates, ORF counts, model ranges) must be identical.  Null and bias scores are fp32
re-statements of the same arithmetic (1e-4).  The frameshift Forward score carries the tolerance of
tests/test_frameshift_gpu.py (table log-sum accumulated in a different association: 1e-4 relative + 5e-3 nats), and the
P-values derived from it inherit that as a relative factor exp(lambda * delta / ln 2)."""
import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["strict", "fast"])
def fs_mode(request, gpu_ctx):
    """Every test of this file runs in both modes of the frameshift recursions: strict (the library's default; the exact
    comparisons of tests/test_fs_strict_gpu.py apply to it as well) and fast (wavefront scans), whose scores carry the
    tolerances stated below."""
    gpu_ctx.set_fs_strict(request.param == "strict")
    yield request.param
    gpu_ctx.set_fs_strict(True)


def frameshifted_windows(rng, model, n=40, L_flank=250):
    """Planted genes, most of them with 1-3 single-nucleotide insertions or deletions, on either strand."""
    wins = []
    genes = common.emit_from_model(rng, model, n // 2, flank=5) + common.emit_from_model(rng, model, n - n // 2, flank=5, sharpen=2.0)
    for i, aa in enumerate(genes):
        nt = list(common.revtranslate(rng, aa, model.basic))
        for _ in range(int(rng.integers(0, 4))):
            p = int(rng.integers(10, max(11, len(nt) - 10)))
            if rng.random() < 0.5:
                del nt[p]
            else:
                nt.insert(p, int(rng.integers(0, 4)))
        pre = rng.integers(0, 4, size=int(rng.integers(0, L_flank)))
        post = rng.integers(0, 4, size=int(rng.integers(0, L_flank)))
        w = np.concatenate([pre, np.array(nt, dtype=np.int64), post]).astype(np.uint8)
        if i % 2:
            w = (3 - w[::-1]).astype(np.uint8)
        wins.append(w)
    # two genes in one long window (two DNA windows, or one merged), and background
    a = np.concatenate([wins[0], rng.integers(0, 4, size=4000).astype(np.uint8), wins[2]])
    b = np.concatenate([wins[1], rng.integers(0, 4, size=60).astype(np.uint8), wins[3]])
    return wins + [a, b] + common.random_dna(rng, 20, 1000)


def run_both(ctx, path, idx, wins):
    model = ol.Model(path, idx)
    hmm = ba.HMM(path, idx)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    stats, res, fw = pipe.run_frameshift(om3, ba.SeqBlock(ctx, wins))
    pli, ores, per_seq, ofw, per_seq_w = model.run_pipeline_fs(wins)
    return model, stats, res, fw, pli, ores, per_seq, ofw, per_seq_w


def compare(model, stats, fw, pli, ofw, per_seq_w, F3=1e-5):
    want = []
    for w, (a, b) in enumerate(per_seq_w):
        want += [(w, o) for o in ofw[a:b]]
    want.sort(key=lambda t: (t[0], t[1].strand, t[1].n))
    got = sorted(fw, key=lambda g: (g.window, g.strand, g.n))
    assert len(got) == len(want)
    lam = model.om.contents.evparam[5]
    unsure = 0
    pos = 0
    for g, (w, o) in zip(got, want):
        assert (g.window, g.strand, g.n, g.length, g.orf_cnt, g.k_min, g.k_max) == (w, o.strand, o.n, o.length, o.orf_cnt, o.k_min, o.k_max)
        assert abs(g.tot_orfsc - o.tot_orfsc) <= 2e-3 + 1e-4 * abs(o.tot_orfsc)       # sum of Forward scores, each 1e-4 relative
        assert abs(g.nullsc - o.nullsc) <= 1e-5 * max(1.0, abs(o.nullsc))
        assert abs(g.filtersc - o.filtersc) <= 1e-4 * max(1.0, abs(o.filtersc))
        tol = 5e-3 + 1e-4 * abs(o.fwdsc)
        assert abs(g.fwdsc - o.fwdsc) <= tol
        fac = np.exp(lam * (tol + 1e-3) / np.log(2.0)) * 1.001
        for a, b in ((g.P_fs, o.P_fs), (g.P_null, o.P_null), (g.P_tot, o.P_tot)):
            assert b / fac <= a <= b * fac or (a < 1e-300 and b < 1e-300)
        # the branch must agree unless the oracle's own decision is within the score tolerance of flipping
        clear = (o.P_fs > F3 * fac or o.P_fs < F3 / fac) and (o.P_null > o.P_tot * fac * fac or o.P_null < o.P_tot / (fac * fac))
        if clear:
            assert g.branch == o.branch
        else:
            unsure += 1
    assert unsure <= max(2, len(got) // 10)
    if unsure == 0:
        assert stats.pos_past_fwd == pli.pos_past_fwd
    for name in ("nres", "n_orfs", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd", "pos_past_msv", "pos_past_bias", "pos_past_vit"):
        assert getattr(stats, name) == getattr(pli, name), name
    return len(got)


def test_recorded_fs_run_on_gpu(gpu_ctx):
    """tutorial/AMP_N-fs.out: 'Residues passing Fwd filter: 411' comes out of the frameshift stage."""
    path = ol.GOLDEN + "/AMP_N.bhmm"
    seqs = [ba.digitize(s, ba.DNA_SYMS) for _, s in ol.read_fasta(ol.GOLDEN + "/target-AMP_N.fa")]
    model, stats, res, fw, pli, ores, per_seq, ofw, per_seq_w = run_both(gpu_ctx, path, 0, seqs)
    assert (stats.nres, stats.pos_past_msv, stats.pos_past_bias, stats.pos_past_vit, stats.pos_past_fwd) == (822, 537, 537, 393, 411)
    assert len(fw) == 1 and (fw[0].n, fw[0].length, fw[0].orf_cnt, fw[0].branch) == (1, 411, 3, 1)
    compare(model, stats, fw, pli, ofw, per_seq_w)


@pytest.mark.parametrize("fasta", ["2OG-FeII_Oxy_3-nt-fs.fa", "2OG-FeII_Oxy_3-nt.fa"])
def test_reference_fs_example(gpu_ctx, fasta):
    """BASELINE configs[0]: testsuite/2OG-FeII_Oxy_3.bhmm against its frameshifted / unshifted DNA targets."""
    path = ol.GOLDEN + "/2OG-FeII_Oxy_3.bhmm"
    seqs = [ba.digitize(s, ba.DNA_SYMS) for _, s in ol.read_fasta(ol.GOLDEN + "/" + fasta)]
    model, stats, res, fw, pli, ores, per_seq, ofw, per_seq_w = run_both(gpu_ctx, path, 0, seqs)
    assert compare(model, stats, fw, pli, ofw, per_seq_w) == 10


@pytest.mark.parametrize("name", ["Caudal_act.bhmm", "PTH2.bhmm", "2OG-FeII_Oxy_3.bhmm"])
def test_planted_frameshifted_genes(gpu_ctx, name):
    rng = np.random.default_rng(31)
    path = ol.GOLDEN + "/" + name
    wins = frameshifted_windows(rng, ol.Model(path, 0))
    model, stats, res, fw, pli, ores, per_seq, ofw, per_seq_w = run_both(gpu_ctx, path, 0, wins)
    n = compare(model, stats, fw, pli, ofw, per_seq_w)
    assert n >= 20 and any(w.branch == 1 for w in fw) and any(w.branch == 2 for w in fw)
    assert any(w.strand == 1 for w in fw) and any(w.orf_cnt > 1 for w in fw)


def compare_domains(model, gdm, odm, per_d, nclustered=0):
    """Domain by domain.  Three grades of agreement, because three kinds of arithmetic feed the envelopes:

    * exact: alignment end points, model range, shifted-codon count and envelope identical, scores at the tolerances of
      tests/test_frameshift_gpu.py.  This is the rule; everything below is bounded.
    * one-step envelope: envelope ends come from thresholds on posterior sums of the 3-codon parsers (rt2 = 0.10,
      p7_domaindef.c:355-372), whose table log-sum arithmetic is only tolerance-equal between the two implementations, so an
      end may fall one step (<= 3 nt) to either side in a borderline case; the alignment may then pick up or drop a weak
      segment at that end.  At most 10% of the domains (and at least 1).
    * clustered regions (<nclustered> of them): their envelopes are consensus end points of 200 stochastic tracebacks
      through the region's Forward matrix.  Two tolerance-equal matrices do not give the same 200 samples, so these envelopes
      agree only statistically: same place (ends within 60 nt -- an end is the widest one that 2% of the samples support --,
      alignments overlapping), score within 2.5 bits.  At most 3 such
      domains per clustered region; a domain too weak to be reported may be missing on one side."""
    want = []
    for w, (a, b) in enumerate(per_d):
        want += [(w, o) for o in odm[a:b]]
    key = lambda w, d: (w, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm, d.n_shifted_codons)
    lam = model.om.contents.evparam[5]
    omap = {}
    for w, o in want:
        omap.setdefault(key(w, o), []).append(o)
    rest_g = []
    resampled = 0
    for g in gdm:
        lst = omap.get(key(g.window, g))
        if not lst:
            rest_g.append(g)
            continue
        o = lst.pop()
        assert abs(g.envsc - o.envsc) <= 5e-3 + 1e-4 * abs(o.envsc)                  # table log-sum association, as test_frameshift_gpu.py
        assert abs(g.oasc - o.oasc) <= 2e-2 + 1e-3 * abs(o.oasc)
        n2tol = 2e-2 + 5e-3 * abs(o.domcorrection)
        if abs(g.domcorrection - o.domcorrection) > n2tol:      # standard-branch domain of a clustered region: the same envelope
            resampled += 1                                       # from a differently sampled ensemble (tests/test_hits_gpu.py)
            n2tol = 0.5
        assert abs(g.domcorrection - o.domcorrection) <= n2tol
        assert abs(g.bitscore - o.bitscore) <= 0.05 + n2tol / np.log(2.0) and abs(g.pre_score - o.pre_score) <= 0.05     # bits; the bias term moves with the correction
        assert abs(g.lnP - o.lnP) <= (0.05 + n2tol / np.log(2.0)) * lam + 1e-6
    rest_o = [(w, o) for w, o in want if any(o is x for x in omap.get(key(w, o), []))]
    shifted = 0
    near = resampled

    def span(d):
        return min(d.ienv, d.jenv), max(d.ienv, d.jenv)
    for g in rest_g:
        best = None
        for idx, (w, o) in enumerate(rest_o):
            if w != g.window:
                continue
            lo, hi = max(span(g)[0], span(o)[0]), min(span(g)[1], span(o)[1])
            if hi - lo + 1 >= 0.5 * (span(o)[1] - span(o)[0] + 1) and (best is None or hi - lo > best[0]):
                best = (hi - lo, idx)
        if best is None:
            assert not g.reported or g.bitscore < 12.0, "GPU-only domain"         # a weak extra cluster
            near += 1
            continue
        w, o = rest_o.pop(best[1])
        if abs(g.ienv - o.ienv) <= 3 and abs(g.jenv - o.jenv) <= 3 and abs(g.envsc - o.envsc) <= 0.1 + 1e-4 * abs(o.envsc) and abs(g.bitscore - o.bitscore) <= 0.5:
            shifted += 1
        else:
            assert abs(g.ienv - o.ienv) <= 60 and abs(g.jenv - o.jenv) <= 60 and abs(g.bitscore - o.bitscore) <= 2.5
            near += 1
    for w, o in rest_o:
        assert not o.reported or o.bitscore < 12.0, "oracle-only domain"
        near += 1
    assert near <= 3 * nclustered
    assert shifted <= max(1, len(want) // 10) + (nclustered if near < 3 * nclustered else 0)
    return len(gdm)


def test_recorded_fs_hit_on_gpu(gpu_ctx):
    """tutorial/AMP_N-fs.tbl: score 82.8, bias 0.1, hmm 1..131, ali 1..402, 6 shifted codons -- from the GPU path."""
    path = ol.GOLDEN + "/AMP_N.bhmm"
    seqs = [ba.digitize(s, ba.DNA_SYMS) for _, s in ol.read_fasta(ol.GOLDEN + "/target-AMP_N.fa")]
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(gpu_ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(gpu_ctx, seqs))
    assert len(fw) == 1 and len(dm) == 1 and nskip == 0
    d = dm[0]
    assert (d.ihmm, d.jhmm, d.iali, d.jali, d.reported, d.n_shifted_codons) == (1, 131, 1, 402, 1, 6)
    assert "%.1f" % d.bitscore == "82.8" and "%.1f" % (d.dombias / np.log(2.0)) == "0.1"
    model = ol.Model(path, 0)
    _, _, _, odm, per_d, _ = model.run_pipeline_fsdom(seqs)
    compare_domains(model, dm, odm, per_d)


@pytest.mark.parametrize("name", ["2OG-FeII_Oxy_3.bhmm", "PTH2.bhmm"])
def test_domains_of_planted_frameshifted_genes(gpu_ctx, name):
    rng = np.random.default_rng(41)
    path = ol.GOLDEN + "/" + name
    model = ol.Model(path, 0)
    wins = frameshifted_windows(rng, model, n=24)
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(gpu_ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(gpu_ctx, wins))
    _, ofw, per_w, odm, per_d, oskip = model.run_pipeline_fsdom(wins)
    assert sorted((w.window, w.strand, w.n, w.branch) for w in fw) == sorted((i, o.strand, o.n, o.branch) for i, (a, b) in enumerate(per_w) for o in ofw[a:b])
    assert nskip == oskip
    n = compare_domains(model, dm, odm, per_d, nskip)
    assert n >= 3 and any(d.n_shifted_codons > 0 for d in dm) and any(d.strand == 1 for d in dm)
    # both branches of p7_pli_Frameshift produce hits: codon-model domains and standard domains on window coordinates
    assert any(fw[d.fs_window].branch == 1 for d in dm) and any(fw[d.fs_window].branch == 2 for d in dm)


def test_negative_alignment_score_drops_the_domain(gpu_ctx):
    """p7_pli_computeAliScores_BATH + p7_domaindef.c:1072,1286 in the --fs pipeline: mutated genes (25% of the residues
    replaced) with nucleotide insertions and deletions; the oracle's counter shows the rule fired on this input."""
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    model = ol.Model(path, 0)
    rng = np.random.default_rng(1)
    wins = []
    for aa in common.emit_from_model(rng, model, 150, flank=5):
        aa = [a if rng.random() > 0.25 else int(rng.integers(0, 20)) for a in aa]
        nt = list(common.revtranslate(rng, aa, model.basic))
        for _ in range(int(rng.integers(1, 4))):
            p = int(rng.integers(10, max(11, len(nt) - 10)))
            if rng.random() < 0.5:
                del nt[p]
            else:
                nt.insert(p, int(rng.integers(0, 4)))
        wins.append(np.array(nt, dtype=np.uint8))
    before = ol.aliscore_drops()
    _, ofw, per_w, odm, per_d, oskip = model.run_pipeline_fsdom(wins)
    assert ol.aliscore_drops() - before >= 1
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(gpu_ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(gpu_ctx, wins))
    assert nskip == oskip
    assert compare_domains(model, dm, odm, per_d, nskip) >= 8
    assert len(dm) == sum(b - a for a, b in per_d)            # no extra domain hiding in compare_domains' allowances


@pytest.mark.parametrize("M", [700, 1024])
def test_frameshift_path_with_a_long_model(gpu_ctx, tmp_path, M):
    """A 700-node synthetic model (11 nodes per lane in the frameshift kernels -> the 12-node instantiation, 5.7 MB of
    5-codon emissions) and the 1024-node model of BASELINE configs[4] (16 nodes per lane, the kernels' largest
    instantiation): the whole --fs path to hits against the oracle."""
    path = common.write_synthetic_bhmm(str(tmp_path / ("s%d.bhmm" % M)), M, seed=M)
    model = ol.Model(path, 0)
    rng = np.random.default_rng(12)
    wins = frameshifted_windows(rng, model, n=8, L_flank=60)[:10] + common.random_dna(rng, 4, 1200)
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(gpu_ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(gpu_ctx, wins))
    _, ofw, per_w, odm, per_d, oskip = model.run_pipeline_fsdom(wins)
    assert sorted((w.window, w.strand, w.n, w.length) for w in fw) == sorted((i, o.strand, o.n, o.length) for i, (a, b) in enumerate(per_w) for o in ofw[a:b])
    assert nskip == oskip
    assert compare_domains(model, dm, odm, per_d, nskip) >= 3


def test_envelope_batches_give_the_same_hits(gpu_ctx, monkeypatch):
    """The envelope kernels work through the envelopes in batches bounded by the memory of their matrices; a tiny bound
    (many batches) must give exactly what one batch gives."""
    rng = np.random.default_rng(77)
    path = ol.GOLDEN + "/PTH2.bhmm"
    model = ol.Model(path, 0)
    wins = frameshifted_windows(rng, model, n=30)
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(gpu_ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    out = []
    for mb in ("100000", "1"):                       # one batch; about one envelope per batch
        monkeypatch.setenv("BATH_HIP_ENV_MB", mb)
        _, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(gpu_ctx, wins))
        out.append((nskip, [(d.window, d.fs_window, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm, d.envsc, d.oasc, d.domcorrection, d.bitscore, d.lnP,
                             d.n_shifted_codons, d.n_stops, d.pid, d.cigar) for d in dm]))
    assert out[0] == out[1] and len(out[0][1]) >= 10


def test_cascade_lanes_give_the_same_frameshift_hits(gpu_ctx, monkeypatch):
    """Large blocks run the cascade as two concurrent parts (lanes), each selecting its F4 survivors on its own device state;
    candidate ids, window numbers and residue addresses are merged over the lanes.  One lane and two lanes must give exactly
    the same DNA windows, branch decisions and hits."""
    rng = np.random.default_rng(78)
    path = ol.GOLDEN + "/PTH2.bhmm"
    model = ol.Model(path, 0)
    wins = frameshifted_windows(rng, model, n=30)
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(gpu_ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    out = []
    for lanes in ("1", "2", "3"):
        monkeypatch.setenv("BATH_HIP_LANES", lanes)
        st, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(gpu_ctx, wins))
        out.append(([getattr(st, k) for k in ("nres", "n_orfs", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd", "pos_past_fwd")], nskip,
                    [(w.window, w.strand, w.n, w.length, w.orf_cnt, w.k_min, w.k_max, w.branch, w.fwdsc, w.filtersc, w.P_fs, w.P_tot) for w in fw],
                    [(d.window, d.fs_window, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm, d.envsc, d.oasc, d.domcorrection, d.bitscore, d.lnP,
                      d.n_shifted_codons, d.n_stops, d.pid, d.cigar) for d in dm]))
    assert out[0] == out[1] == out[2]
    assert len(out[0][2]) >= 20 and len(out[0][3]) >= 10
    assert {w[7] for w in out[0][2]} == {1, 2}                      # both branches taken: the standard branch reads the lanes' residue pools


@pytest.mark.parametrize("switch", ["BATH_HIP_FS_STD_SERIAL", "BATH_HIP_FS_REGION_COPY", "BATH_HIP_FS_TWO_BATCHES", "BATH_HIP_FS_LIVE", "BATH_HIP_HOST_THREADS"])
def test_scheduling_switches_do_not_change_the_hits(gpu_ctx, monkeypatch, switch):
    """How the domain stage is scheduled -- the standard branch on its own thread and stream or afterwards, the region Forward
    written straight to page-locked memory or copied, one envelope batch or two, ensembles started per region or after the whole
    region Forward, the number of ensemble threads -- must not change a single hit."""
    rng = np.random.default_rng(79)
    path = ol.GOLDEN + "/PTH2.bhmm"
    model = ol.Model(path, 0)
    wins = frameshifted_windows(rng, model, n=30)
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(gpu_ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    out = []
    for value in (None, "0" if switch == "BATH_HIP_FS_LIVE" else "1"):
        if value is None:
            monkeypatch.delenv(switch, raising=False)
        else:
            monkeypatch.setenv(switch, value)
        _, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(gpu_ctx, wins))
        out.append((nskip, sorted((d.window, d.fs_window, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm, d.envsc, d.oasc, d.domcorrection, d.bitscore, d.lnP,
                                   d.n_shifted_codons, d.n_stops, d.pid, d.cigar) for d in dm)))
    assert out[0] == out[1] and len(out[0][1]) >= 10 and out[0][0] >= 1        # at least one clustered region went through the ensembles


@pytest.mark.parametrize("name", ["Caudal_act.bhmm", "PTH2.bhmm", "2OG-FeII_Oxy_3.bhmm"])
def test_wave_and_lane_frameshift_traceback_agree(gpu_ctx, monkeypatch, name):
    """p7_OATrace_Frameshift, the null2 correction over the aligned residues and the alignment columns of an envelope by the whole
    wave (fs5_trace_wave_kernel: look-ahead along the slope-3 diagonal, flanks 64 rows at a time, a lane per column) and by one lane
    (BATH_HIP_FS_TRACE_LANE=1, the serial restatement): every field of every domain and every trace column -- state, node,
    position, codon length, posterior -- must be identical bit for bit; the two make the same decisions on the same values."""
    rng = np.random.default_rng(131)
    path = ol.GOLDEN + "/" + name
    model = ol.Model(path, 0)
    wins = frameshifted_windows(rng, model, n=36)
    for q in range(0, len(wins), 7):                                      # degenerate nucleotides: the X rule of the columns' scores
        wins[q] = wins[q].copy()
        wins[q][rng.integers(0, len(wins[q]), size=2)] = 4 + rng.integers(0, 11, size=2)
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(gpu_ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    out = []
    for lane in ("0", "1"):
        monkeypatch.setenv("BATH_HIP_FS_TRACE_LANE", lane)
        _, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(gpu_ctx, wins))
        out.append((nskip, dm, pipe.traces()))
    (na, da, ta), (nb, db, tb) = out
    assert na == nb and len(da) == len(db) >= 10
    nfs = nshift = 0
    bits = lambda v: np.float32(v).view(np.uint32)
    for a, b, (t1, st1, k1, i1, c1, pp1), (t2, st2, k2, i2, c2, pp2) in zip(da, db, ta, tb):
        ka = (a.window, a.fs_window, a.ienv, a.jenv, a.iali, a.jali, a.ihmm, a.jhmm, a.n_shifted_codons, a.n_stops, a.pid, a.cigar, a.reported)
        kb = (b.window, b.fs_window, b.ienv, b.jenv, b.iali, b.jali, b.ihmm, b.jhmm, b.n_shifted_codons, b.n_stops, b.pid, b.cigar, b.reported)
        assert ka == kb
        for f in ("envsc", "oasc", "domcorrection", "bitscore", "lnP"):
            assert bits(getattr(a, f)) == bits(getattr(b, f)), (f, getattr(a, f), getattr(b, f))
        assert (t1.N, t1.win_start, t1.frameshift) == (t2.N, t2.win_start, t2.frameshift)
        assert np.array_equal(st1, st2) and np.array_equal(k1, k2) and np.array_equal(i1, i2) and np.array_equal(c1, c2)
        assert np.array_equal(pp1.view(np.uint32), pp2.view(np.uint32))
        if t1.frameshift:
            nfs += 1
            nshift += int((c1[st1 == ba.T_M] != 3).sum())
    assert nfs >= 5 and nshift >= 3                                       # quasi-codons: where the walk leaves the slope-3 diagonal
