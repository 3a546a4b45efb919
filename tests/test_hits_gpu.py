"""The standard branch after the Forward filter on the GPU (bath_hip_pipeline_hits): Backward parser, domain decoding,
region heuristics, the envelope's full Forward/Backward (unihit), posterior decoding, optimal-accuracy fill and traceback,
null2, and the hit's score arithmetic -- against the oracle (oracle/domaindef.c) and against what the reference itself
recorded (tutorial/PTH2.tbl, tutorial/AMP_N.out).

Integer outputs (envelope, alignment and model coordinates) must be identical -- also for multi-domain regions, where both
sides sample 200 stochastic tracebacks from the same random-number stream (oracle/stotrace.c has the caveats).  The envelope score is a Forward score
(1e-4 relative, as in tests/test_filters_gpu.py); oasc / domcorrection are sums of posteriors computed from products of
Forward and Backward values, each 1e-4 relative, over up to Ld terms: 2e-3 absolute + 1e-3 relative.  Bit scores inherit
the envelope score's tolerance divided by ln 2."""
import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol

pytestmark = pytest.mark.gpu

RECORDED_STD_HITS = {   # tutorial/PTH2.tbl and tutorial/AMP_N.out: (hmm from, hmm to, ali from, ali to, score, bias)
    "PTH2.bhmm": ("target-PTH2.fa", [(2, 116, 672, 325, "110.6", "0.3"), (35, 116, 1486, 1731, "86.4", "0.0"),
                                     (71, 113, 2468, 2343, "36.2", "0.0"), (2, 30, 1273, 1359, "36.0", "0.3")]),
    "AMP_N.bhmm": ("target-AMP_N.fa", [(None, None, 7, 234, "47.8", "0.0")]),
}


@pytest.fixture(scope="module")
def ctx():
    return ba.Context(0)


def gpu_hits(ctx, path, idx, wins):
    hmm = ba.HMM(path, idx)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
    return pipe.run_hits(ba.SeqBlock(ctx, wins))


@pytest.mark.parametrize("hmmfile", sorted(RECORDED_STD_HITS))
def test_hits_match_recorded_runs(ctx, hmmfile):
    fasta, want = RECORDED_STD_HITS[hmmfile]
    if hmmfile == "PTH2.bhmm":
        rows = [l.split() for l in open(ol.GOLDEN + "/PTH2.tbl") if l and l[0] != "#"]
        assert [(int(r[6]), int(r[7]), int(r[9]), int(r[10]), r[12], r[13]) for r in rows] == want
    seqs = [ol.digitize_dna(s) for _, s in ol.read_fasta(ol.GOLDEN + "/" + fasta)]
    stats, dm, nskip = gpu_hits(ctx, ol.GOLDEN + "/" + hmmfile, 0, seqs)
    got = sorted(dm, key=lambda d: -d.bitscore)
    assert nskip == 0 and len(got) == len(want) and all(d.reported for d in got)
    for d, (h1, h2, a1, a2, score, bias) in zip(got, want):
        assert (d.iali, d.jali) == (a1, a2)
        if h1 is not None:
            assert (d.ihmm, d.jhmm) == (h1, h2)
        assert "%.1f" % d.bitscore == score and "%.1f" % (d.dombias / np.log(2.0)) == bias


LEDGER = []


def _write_ledger():
    """How the standard branch's domains compared in this run: gpurun_out/std_branch_ledger.json (every compare_hits call)."""
    import json, os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        tot = {k: sum(e[k] for e in LEDGER) for k in ("domains", "clustered_regions", "exact", "unmatched_gpu", "unmatched_oracle", "same_envelope_other_ensemble")}
        json.dump({"total": tot, "calls": LEDGER}, open(os.path.join(out, "std_branch_ledger.json"), "w"), indent=1)
    except OSError:
        pass


def compare_hits(dm, odm, per_d, nskip, onskip):
    """Every domain must have its exact counterpart (coordinates identical, scores at the stated tolerances) -- including the
    domains of clustered regions (nskip of them), which come from 200 sampled tracebacks through a Forward matrix that agrees
    with the oracle's to rounding only: a sample COULD take a different turn on the two sides and change that region's
    envelopes.  Rounds 1-3 allowed up to 3 such domains per clustered region; a ledger of every comparison
    (gpurun_out/std_branch_ledger.json: round 4, 278 domains, 45 clustered regions over the GPU tier's inputs, plus the bench's
    samples) shows it never happens on any input in this tree, so the allowance is gone: an unmatched domain fails the test and
    the ledger entry says which one (window, envelope, score)."""
    want = []
    for w, (a, b) in enumerate(per_d):
        want += [(w, o) for o in odm[a:b]]
    assert nskip == onskip
    key = lambda w, d: (w, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm)
    omap = {}
    for w, o in want:
        omap.setdefault(key(w, o), []).append(o)
    rest_g = []
    resampled = 0
    for g in dm:
        lst = omap.get(key(g.window, g))
        if not lst:
            rest_g.append(g)
            continue
        o = lst.pop()
        assert g.strand == (1 if o.ienv > o.jenv else 0)
        assert abs(g.envsc - o.envsc) <= 1e-4 * max(1.0, abs(o.envsc))
        # the early E-value test with the running residue count of the hit's window and strand (p7_pipeline.c:1246); a P-value within
        # its tolerance of the threshold could fall on either side
        assert g.reported == o.reported or abs(g.lnP - o.lnP) > 0.0, (g.window, g.lnP, o.lnP)
        assert abs(g.oasc - o.oasc) <= 2e-3 + 1e-3 * abs(o.oasc)
        n2tol = 2e-3 + 1e-3 * abs(o.domcorrection)
        if abs(g.domcorrection - o.domcorrection) > n2tol:
            # same envelope from a differently sampled ensemble: the correction is a mean over 200 sampled traces
            resampled += 1
            n2tol = 0.5
        assert abs(g.domcorrection - o.domcorrection) <= n2tol
        assert abs(g.dombias - o.dombias) <= n2tol
        assert abs(g.bitscore - o.bitscore) <= (1e-4 * max(1.0, abs(o.envsc)) + 2e-3 + n2tol) / np.log(2.0)
        assert abs(g.pre_score - o.pre_score) <= (1e-4 * max(1.0, abs(o.envsc)) + 1e-3) / np.log(2.0)
        assert abs(g.lnP - o.lnP) <= 0.8 * ((1e-4 * max(1.0, abs(o.envsc)) + 2e-3 + n2tol) / np.log(2.0)) + 1e-6
    rest_o = [(w, o) for w, o in want if any(o is x for x in omap.get(key(w, o), []))]
    LEDGER.append({"domains": len(dm), "clustered_regions": int(nskip), "exact": len(dm) - len(rest_g), "unmatched_gpu": len(rest_g), "unmatched_oracle": len(rest_o),
                   "same_envelope_other_ensemble": resampled,
                   "unmatched": [{"window": int(g.window), "env": [int(g.ienv), int(g.jenv)], "bits": round(float(g.bitscore), 2)} for g in rest_g]})
    _write_ledger()
    assert not rest_g and not rest_o and resampled == 0, LEDGER[-1]
    return len(dm)


@pytest.mark.parametrize("hmmfile,idx", [("PTH2.bhmm", 0), ("Caudal_act.bhmm", 0), ("AMP_N.bhmm", 0), ("MET-ct4.bhmm", 1)])
def test_hits_match_oracle_on_planted_genes(ctx, hmmfile, idx):
    path = ol.GOLDEN + "/" + hmmfile
    model = ol.Model(path, idx)
    rng = np.random.default_rng(77 + idx)
    wins = []
    genes = common.emit_from_model(rng, model, 16, flank=5) + common.emit_from_model(rng, model, 16, flank=5, sharpen=2.0)
    for i, aa in enumerate(genes):
        nt = np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8)
        pre = rng.integers(0, 4, size=int(rng.integers(0, 300))).astype(np.uint8)
        post = rng.integers(0, 4, size=int(rng.integers(0, 300))).astype(np.uint8)
        w = np.concatenate([pre, nt, post]).astype(np.uint8)
        if i % 2:
            w = (3 - w[::-1]).astype(np.uint8)
        wins.append(w)
    # truncated genes (short envelopes, alignments at the ORF's ends) and background
    for i in range(6):
        g = np.array(common.revtranslate(rng, genes[i], model.basic), dtype=np.uint8)
        cut = max(60, len(g) // 3)
        wins.append(g[:cut] if i % 2 else g[-cut:])
    # two and three genes in one reading frame without a stop between them: multi-domain regions, resolved by clustering an
    # ensemble of stochastic tracebacks (p7_domaindef.c:539-583); nskip counts the regions that went that way
    for i in range(0, 12, 2):
        tandem = list(genes[i]) + list(genes[i + 1]) + (list(genes[i + 2]) if i % 4 == 0 else [])
        nt = np.array(common.revtranslate(rng, tandem, model.basic), dtype=np.uint8)
        w = np.concatenate([rng.integers(0, 4, size=33).astype(np.uint8), nt, rng.integers(0, 4, size=60).astype(np.uint8)])
        wins.append((3 - w[::-1]).astype(np.uint8) if i % 4 else w)
    wins += common.random_dna(rng, 30, 1000)
    stats, dm, nskip = gpu_hits(ctx, path, idx, wins)
    pli, odm, per_d, onskip = model.run_pipeline_hits(wins)
    assert (stats.n_past_fwd, stats.pos_past_fwd) == (pli.n_past_fwd, pli.pos_past_fwd)
    n = compare_hits(dm, odm, per_d, nskip, onskip)
    assert n >= 8 and nskip >= 2
    # a clustered region gives several hits on one ORF: some window must carry more than one hit on a strand
    per = {}
    for d in dm:
        per[(d.window, d.strand)] = per.get((d.window, d.strand), 0) + 1
    assert max(per.values()) >= 2


def test_negative_alignment_score_drops_the_domain(ctx):
    """p7_pli_computeAliScores_BATH + p7_domaindef.c:1286: an envelope whose optimal-accuracy alignment sums to a negative
    per-position score gives no domain.  Heavily mutated genes (30% of the residues replaced) produce such envelopes now and
    then; some windows also carry degenerate nucleotides inside aligned codons (those columns score as X).  The oracle's
    counter shows that the rule fired; the hit lists must agree as everywhere else."""
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    model = ol.Model(path, 0)
    rng = np.random.default_rng(1)
    wins = []
    for aa in common.emit_from_model(rng, model, 300, flank=5):
        aa = [a if rng.random() > 0.3 else int(rng.integers(0, 20)) for a in aa]
        wins.append(np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8))
    rng2 = np.random.default_rng(2)
    for aa in common.emit_from_model(rng2, model, 24, flank=5):
        nt = np.array(common.revtranslate(rng2, aa, model.basic), dtype=np.uint8)
        for p in rng2.integers(0, len(nt), size=4):
            nt[int(p)] = 15                                     # N
        rc = np.where(nt < 4, 3 - nt, nt)[::-1].astype(np.uint8)          # reverse complement; N stays N
        wins.append(nt if len(wins) % 2 else rc)
    before = ol.aliscore_drops()
    pli, odm, per_d, onskip = model.run_pipeline_hits(wins)
    assert ol.aliscore_drops() - before >= 1
    stats, dm, nskip = gpu_hits(ctx, path, 0, wins)
    assert compare_hits(dm, odm, per_d, nskip, onskip) >= 20
    # compare_hits tolerates a few unmatched domains per clustered region; on this input none is needed, so a domain the rule
    # should have dropped cannot hide there
    key = lambda w, d: (w, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm)
    assert sorted(key(d.window, d) for d in dm) == sorted(key(w, o) for w, (a, b) in enumerate(per_d) for o in odm[a:b])


def test_hits_empty_and_background(ctx):
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    rng = np.random.default_rng(5)
    wins = common.random_dna(rng, 50, 1000) + [np.zeros(0, dtype=np.uint8), np.zeros(10, dtype=np.uint8)]
    stats, dm, nskip = gpu_hits(ctx, path, 0, wins)
    model = ol.Model(path, 0)
    pli, odm, per_d, onskip = model.run_pipeline_hits(wins)
    compare_hits(dm, odm, per_d, nskip, onskip)


def test_hits_of_an_empty_block(ctx):
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    stats, dm, nskip = gpu_hits(ctx, path, 0, [])
    assert (stats.nres, stats.n_orfs, len(dm), nskip) == (0, 0, 0, 0)
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    st, fw, dm, nskip = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct).run_frameshift_domains(om3, om5, ba.SeqBlock(ctx, []))
    assert (st.nres, len(fw), len(dm), nskip) == (0, 0, 0, 0)


def test_wave_and_lane_envelope_kernels_agree(ctx, monkeypatch):
    """Decoding, optimal-accuracy fill and null2 of an envelope run with a wave per envelope (rows in registers, the D chain
    a wavefront scan); the lane-per-envelope kernel does the same serially (BATH_HIP_STD_SERIAL=1).  Coordinates, oasc and
    the envelope score must be identical (max / select arithmetic); the null2 correction differs only by the association of
    one sum."""
    path = ol.GOLDEN + "/MET-ct4.bhmm"                   # M = 409: 7 nodes per lane -> the 8-node instantiation
    model = ol.Model(path, 0)
    rng = np.random.default_rng(99)
    wins = []
    for i, aa in enumerate(common.emit_from_model(rng, model, 24, flank=4, sharpen=2.0)):
        nt = np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8)
        w = np.concatenate([rng.integers(0, 4, size=60).astype(np.uint8), nt, rng.integers(0, 4, size=45).astype(np.uint8)])
        wins.append((3 - w[::-1]).astype(np.uint8) if i % 2 else w)
    out = []
    # the block-of-four-waves fill (the default from 193 nodes on), the serial kernel, the one-wave fill
    for serial, mw in (("0", None), ("1", None), ("0", "0")):
        monkeypatch.setenv("BATH_HIP_STD_SERIAL", serial)
        if mw is None:
            monkeypatch.delenv("BATH_HIP_STD_FILL_MW", raising=False)
        else:
            monkeypatch.setenv("BATH_HIP_STD_FILL_MW", mw)
        _, dm, nskip = gpu_hits(ctx, path, 0, wins)
        out.append((nskip, sorted((d.window, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm, d.envsc, d.oasc, d.ali_columns, d.pid, d.cigar, d.domcorrection) for d in dm)))
    for other in (1, 2):
        assert out[0][0] == out[other][0] and len(out[0][1]) == len(out[other][1]) >= 15
        for a, b in zip(out[0][1], out[other][1]):
            assert a[:12] == b[:12]
            assert abs(a[12] - b[12]) <= 1e-4 * max(1.0, abs(b[12]))


@pytest.mark.parametrize("hmmfile", ["MET-ct4.bhmm", "PTH2.bhmm", "Caudal_act.bhmm"])
def test_wave_and_lane_traceback_agree(ctx, monkeypatch, hmmfile):
    """The optimal-accuracy traceback of an envelope, its alignment score and the posteriors of its columns by the whole wave
    (std_trace_wave_kernel: look-ahead along the diagonal, the C flank 64 rows at a time, a lane per column afterwards) and by
    one lane (BATH_HIP_STD_TRACE_LANE=1, the serial restatement of p7_OATrace): every field of every domain and every trace
    column must be identical, bit for bit -- the two make the same decisions on the same values."""
    path = ol.GOLDEN + "/" + hmmfile
    model = ol.Model(path, 0)
    rng = np.random.default_rng(7)
    wins = []
    for i, aa in enumerate(common.emit_from_model(rng, model, 40, flank=6, sharpen=1.3)):      # a soft model: indels and ragged ends
        nt = np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8)
        w = np.concatenate([rng.integers(0, 4, size=int(rng.integers(3, 400))).astype(np.uint8), nt, rng.integers(0, 4, size=int(rng.integers(0, 300))).astype(np.uint8)])
        if i % 5 == 0:
            w[rng.integers(0, len(w), size=3)] = 4 + rng.integers(0, 11, size=3)               # degenerate nucleotides: the X rule of the alignment score
        wins.append((3 - w[::-1]).astype(np.uint8) if i % 2 and w.max() < 4 else w)
    out = []
    for lane in ("0", "1"):
        monkeypatch.setenv("BATH_HIP_STD_TRACE_LANE", lane)
        hmm = ba.HMM(path, 0)
        om = ba.OProfile(ctx, ba.Profile(hmm))
        pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
        _, dm, nskip = pipe.run_hits(ba.SeqBlock(ctx, wins))
        tr = pipe.traces()
        out.append((nskip, dm, tr))
    (na, da, ta), (nb, db, tb) = out
    assert na == nb and len(da) == len(db) >= 10
    fields = ("window", "ienv", "jenv", "iali", "jali", "ihmm", "jhmm", "envsc", "oasc", "ali_columns", "pid", "cigar", "domcorrection", "bitscore", "lnP", "reported", "stops", "shifts")
    kinds = set()
    for a, b, (t1, st1, k1, i1, c1, pp1), (t2, st2, k2, i2, c2, pp2) in zip(da, db, ta, tb):
        for f in fields:
            if hasattr(a, f):
                va, vb = getattr(a, f), getattr(b, f)
                assert va == vb or (isinstance(va, float) and np.float32(va).view(np.uint32) == np.float32(vb).view(np.uint32)), (f, va, vb)
        assert (t1.N, t1.win_start, t1.orf_start) == (t2.N, t2.win_start, t2.orf_start)
        assert np.array_equal(st1, st2) and np.array_equal(k1, k2) and np.array_equal(i1, i2) and np.array_equal(pp1.view(np.uint32), pp2.view(np.uint32))
        kinds |= set(int(x) for x in st1)
    assert ba.T_M in kinds and (hmmfile != "MET-ct4.bhmm" or len(kinds) >= 2)


@pytest.mark.parametrize("hmmfile", ["PTH2.bhmm", "MET-ct4.bhmm"])
def test_kept_forward_rows_give_the_same_hits(ctx, monkeypatch, hmmfile):
    """The domain stage reads the Forward parser's special-state rows of the ORFs that passed the Forward filter; the cascade's own
    Forward launch leaves them (the reference keeps pli->oxf from the filter to p7_domaindef) and the domain stage copies the
    survivors' rows instead of running the parser again.  BATH_HIP_KEEP_FWD=0 runs it again: every field of every domain must be
    identical, bit for bit."""
    path = ol.GOLDEN + "/" + hmmfile
    model = ol.Model(path, 0)
    rng = np.random.default_rng(23)
    wins = []
    for i, aa in enumerate(common.emit_from_model(rng, model, 30, flank=5, sharpen=1.5)):
        nt = np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8)
        w = np.concatenate([rng.integers(0, 4, size=int(rng.integers(3, 300))).astype(np.uint8), nt, rng.integers(0, 4, size=int(rng.integers(0, 300))).astype(np.uint8)])
        wins.append((3 - w[::-1]).astype(np.uint8) if i % 2 else w)
    out = []
    for keep in ("1", "0"):
        monkeypatch.setenv("BATH_HIP_KEEP_FWD", keep)
        _, dm, nskip = gpu_hits(ctx, path, 0, wins)
        out.append((nskip, [(d.window, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm, np.float32(d.envsc).view(np.uint32), np.float32(d.oasc).view(np.uint32),
                             np.float32(d.domcorrection).view(np.uint32), np.float32(d.bitscore).view(np.uint32), d.lnP, d.ali_columns, d.pid, d.cigar, d.reported) for d in dm]))
    assert out[0] == out[1] and len(out[0][1]) >= 15


def test_hits_with_a_1024_node_model(ctx, tmp_path):
    """BASELINE config 5 shape: 16 nodes per lane in the wave-per-envelope kernels, 4 lanes per ORF in SSV."""
    path = common.write_synthetic_bhmm(str(tmp_path / "s1024.bhmm"), 1024, seed=1024)
    model = ol.Model(path, 0)
    rng = np.random.default_rng(4)
    wins = []
    for i, aa in enumerate(common.emit_from_model(rng, model, 6, flank=3, sharpen=2.0)):
        nt = np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8)
        w = np.concatenate([rng.integers(0, 4, size=45).astype(np.uint8), nt, rng.integers(0, 4, size=30).astype(np.uint8)])
        wins.append((3 - w[::-1]).astype(np.uint8) if i % 2 else w)
    wins += common.random_dna(rng, 6, 1500)
    stats, dm, nskip = gpu_hits(ctx, path, 0, wins)
    pli, odm, per_d, onskip = model.run_pipeline_hits(wins)
    assert (stats.n_past_fwd, stats.pos_past_fwd) == (pli.n_past_fwd, pli.pos_past_fwd)
    assert compare_hits(dm, odm, per_d, nskip, onskip) >= 4


def test_hits_with_degenerate_nucleotides(ctx):
    """IUPAC codes in the DNA (codes 5..15) translate to X: the hit stages see X residues in ORFs, envelopes and alignments
    (null2 of a degenerate residue is the mean over its members; X never equals the consensus)."""
    path = ol.GOLDEN + "/PTH2.bhmm"
    model = ol.Model(path, 0)
    rng = np.random.default_rng(2024)
    wins = []
    for i, aa in enumerate(common.emit_from_model(rng, model, 24, flank=4, sharpen=3.0)):
        nt = np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8)
        w = np.concatenate([rng.integers(0, 4, size=40).astype(np.uint8), nt, rng.integers(0, 4, size=40).astype(np.uint8)])
        m = rng.random(len(w)) < 0.01
        w[m] = rng.choice([5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], size=int(m.sum()))
        wins.append((3 - w[::-1]).astype(np.uint8) if i % 2 and not m.any() else w)
    stats, dm, nskip = gpu_hits(ctx, path, 0, wins)
    pli, odm, per_d, onskip = model.run_pipeline_hits(wins)
    assert (stats.n_orfs, stats.n_past_fwd, stats.pos_past_fwd) == (pli.n_orfs, pli.n_past_fwd, pli.pos_past_fwd)
    assert compare_hits(dm, odm, per_d, nskip, onskip) >= 8


def test_cascade_lanes_give_the_same_hits(ctx, monkeypatch):
    """bath_hip_pipeline_hits on a block cut into concurrent parts (lanes): survivors are selected per lane on the device and
    merged; the hits must be those of a single pass."""
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    model = ol.Model(path, 0)
    rng = np.random.default_rng(101)
    wins = []
    for i, aa in enumerate(common.emit_from_model(rng, model, 30, flank=4, sharpen=2.0)):
        nt = np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8)
        w = np.concatenate([rng.integers(0, 4, size=80).astype(np.uint8), nt, rng.integers(0, 4, size=50).astype(np.uint8)])
        wins.append((3 - w[::-1]).astype(np.uint8) if i % 2 else w)
    wins += common.random_dna(rng, 10, 900)
    out = []
    for lanes in ("1", "2", "3"):
        monkeypatch.setenv("BATH_HIP_LANES", lanes)
        st, dm, nskip = gpu_hits(ctx, path, 0, wins)
        out.append((st.n_past_fwd, st.pos_past_fwd, nskip,
                    sorted((d.window, d.strand, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm, d.envsc, d.oasc, d.ali_columns, d.pid, d.cigar, d.domcorrection, d.bitscore) for d in dm)))
    assert out[0] == out[1] == out[2] and len(out[0][3]) >= 20


def test_cascade_lanes_with_an_empty_part(ctx, monkeypatch):
    """A lane whose part of the block has no ORF at all (stop codons in every frame) contributes nothing and breaks nothing:
    candidate ids, windows and residue addresses of the other lane's survivors are unaffected."""
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    model = ol.Model(path, 0)
    rng = np.random.default_rng(103)
    genes = []
    for i, aa in enumerate(common.emit_from_model(rng, model, 12, flank=4, sharpen=2.0)):
        nt = np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8)
        genes.append(np.concatenate([rng.integers(0, 4, size=40).astype(np.uint8), nt, rng.integers(0, 4, size=40).astype(np.uint8)]))
    stops = [np.tile(np.array([3, 0, 0, 3, 3, 0, 3, 0, 0, 3], dtype=np.uint8), 100) for _ in range(14)]    # TAATTATAAT...: stops in all six frames
    for wins in (genes + stops, stops + genes):
        out = []
        for lanes in ("1", "2"):
            monkeypatch.setenv("BATH_HIP_LANES", lanes)
            st, dm, nskip = gpu_hits(ctx, path, 0, wins)
            out.append((st.n_orfs, st.n_past_fwd, nskip, sorted((d.window, d.strand, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm, d.envsc, d.oasc, d.cigar, d.bitscore) for d in dm)))
        assert out[0] == out[1] and len(out[0][3]) >= 8
