"""CPU tier: the oracle against the reference's own recorded results and unit-test identities.

There are no per-kernel known-answer vectors in the reference tree; what exists is
  (i)  the pipeline counters of recorded bathsearch runs (tutorial/*.out, copied to tests/golden), which pin
       translation + MSV + bias + Viterbi + Forward decisions end to end, and
  (ii) the exact-emulation identities of the reference's unit tests (msvfilter.c:621-658, vitfilter.c:645-685).
"""
import ctypes as C

import numpy as np
import pytest

import common
import oracle_lib as ol

GOLDEN_RUNS = [
    ("PTH2.bhmm", 0, "target-PTH2.fa", (6000, 1503, 1503, 1401, 1287)),
    ("AMP_N.bhmm", 0, "target-AMP_N.fa", (822, 537, 537, 393, 237)),
    ("MET-ct4.bhmm", 0, "target-MET.fa", (71226, 2487, 2298, 666, 549)),
    ("MET-ct4.bhmm", 1, "target-MET.fa", (71226, 9168, 2775, 1062, 618)),
]


def _printed_counters(outfile, which):
    """Parse the which-th 'Internal pipeline statistics summary' block of a recorded bathsearch output."""
    txt = open(ol.GOLDEN + "/" + outfile).read().split("Internal pipeline statistics summary:")[1 + which]
    vals = {}
    for line in txt.splitlines():
        for key, tag in (("nres", "Target sequence(s):"), ("msv", "Residues passing SSV filter:"), ("bias", "Residues passing bias filter:"),
                         ("vit", "Residues passing Vit filter:"), ("fwd", "Residues passing Fwd filter:")):
            if line.startswith(tag):
                rest = line[len(tag):].split()
                vals[key] = int(rest[1].strip("(")) if key == "nres" else int(rest[0])
    return (vals["nres"], vals["msv"], vals["bias"], vals["vit"], vals["fwd"])


@pytest.mark.parametrize("hmmfile,idx,fasta,expect", GOLDEN_RUNS, ids=[g[0] + str(g[1]) for g in GOLDEN_RUNS])
def test_pipeline_counters_match_recorded_runs(hmmfile, idx, fasta, expect):
    outfile = hmmfile.replace(".bhmm", ".out")
    assert _printed_counters(outfile, idx) == expect            # the table above is what the reference printed
    m = ol.Model(ol.GOLDEN + "/" + hmmfile, idx)
    seqs = [ol.digitize_dna(s) for _, s in ol.read_fasta(ol.GOLDEN + "/" + fasta)]
    pli, _, _ = m.run_pipeline(seqs)
    assert (pli.nres, pli.pos_past_msv, pli.pos_past_bias, pli.pos_past_vit, pli.pos_past_fwd) == expect


@pytest.fixture(scope="module")
def caudal():
    return ol.Model(ol.GOLDEN + "/Caudal_act.bhmm")


def test_msv_equals_generic_viterbi_on_same_as_mf(caudal):
    """utest_msv_filter, msvfilter.c:621-658: GViterbi(SameAsMF)/scale_b - 3 == MSVFilter (tol 0.001)."""
    L_ = ol.lib()
    rng = np.random.default_rng(42)
    out = C.c_float()
    n_checked = 0
    for s in common.random_aa(rng, 60, 20, 200, with_degenerate=False):
        L = len(s)
        L_.bo_profile_reconfig_length(caudal.gm, L)
        L_.bo_oprofile_reconfig_length(caudal.om, L)
        gm2 = L_.bo_profile_same_as_mf(caudal.om, caudal.gm)
        d = ol.dsq_from(s)
        st = L_.bo_msvfilter(ol.u8(d), L, caudal.om, C.byref(out)); sc1 = out.value
        L_.bo_gviterbi(ol.u8(d), L, gm2, C.byref(out)); sc2 = out.value / caudal.om.contents.scale_b - 3.0
        L_.bo_profile_free(gm2)
        if st == 0:
            assert abs(sc1 - sc2) < 0.001
            n_checked += 1
    assert n_checked > 40


def test_vitfilter_equals_generic_viterbi_on_same_as_vf(caudal):
    """utest_viterbi_filter, vitfilter.c:645-685."""
    L_ = ol.lib()
    rng = np.random.default_rng(43)
    out = C.c_float()
    seqs = common.random_aa(rng, 40, 20, 200, with_degenerate=False) + common.emit_from_model(rng, caudal, 20)
    for s in seqs:
        L = len(s)
        L_.bo_profile_reconfig_length(caudal.gm, L)
        L_.bo_oprofile_reconfig_length(caudal.om, L)
        gm2 = L_.bo_profile_same_as_vf(caudal.om, caudal.gm)
        d = ol.dsq_from(s)
        st = L_.bo_vitfilter(ol.u8(d), L, caudal.om, C.byref(out)); sc1 = out.value
        L_.bo_gviterbi(ol.u8(d), L, gm2, C.byref(out)); sc2 = out.value / caudal.om.contents.scale_w - 3.0
        L_.bo_profile_free(gm2)
        if st == 0:
            assert abs(sc1 - sc2) < 0.001


def test_viterbi_bath_score_equals_plain(caudal):
    """utest in vitfilter.c:688-760: the window-emitting variant returns the same score; windows lie inside the target/model."""
    L_ = ol.lib()
    rng = np.random.default_rng(44)
    out1, out2 = C.c_float(), C.c_float()
    for s in common.emit_from_model(rng, caudal, 30):
        L = len(s)
        L_.bo_oprofile_reconfig_length(caudal.om, L)
        L_.bo_bg_setlength(C.byref(caudal.bg), L)
        d = ol.dsq_from(s)
        wl = ol.WindowList(); L_.bo_windowlist_init(C.byref(wl))
        fsc = L_.bo_bg_filterscore(C.byref(caudal.bg), ol.u8(d), L)
        st1 = L_.bo_vitfilter(ol.u8(d), L, caudal.om, C.byref(out1))
        st2 = L_.bo_vitfilter_bath(ol.u8(d), L, caudal.om, caudal.sd, fsc, 1e-3, C.byref(wl), C.byref(out2))
        assert st1 == st2 and (out1.value == out2.value or (np.isinf(out1.value) and np.isinf(out2.value)))
        for i in range(wl.count):
            w = wl.w[i]
            assert 1 <= w.n <= L and 1 <= w.k <= caudal.M and 1 <= w.length <= w.k
        L_.bo_windowlist_free(C.byref(wl))


def test_forward_backward_agree(caudal):
    """Fwd == Bwd (fwdback.c unit tests): parser scores agree to fp32 accuracy; Forward >= Viterbi."""
    L_ = ol.lib()
    rng = np.random.default_rng(45)
    f, b, v = C.c_float(), C.c_float(), C.c_float()
    for s in common.random_aa(rng, 10, 30, 300, False) + common.emit_from_model(rng, caudal, 10):
        L = len(s)
        L_.bo_oprofile_reconfig_length(caudal.om, L)
        d = ol.dsq_from(s)
        fx = np.zeros((L + 1) * 6, np.float32); bx = np.zeros((L + 1) * 6, np.float32)
        assert L_.bo_forward_parser(ol.u8(d), L, caudal.om, ol.f32(fx), C.byref(f)) == 0
        assert L_.bo_backward_parser(ol.u8(d), L, caudal.om, ol.f32(fx), ol.f32(bx), C.byref(b)) == 0
        assert abs(f.value - b.value) < 1e-3 + 1e-5 * abs(f.value)
        if L_.bo_vitfilter(ol.u8(d), L, caudal.om, C.byref(v)) == 0:
            assert f.value >= v.value - 3.1      # VF carries the -3 nat NN/CC/JJ approximation


def test_logsum_table():
    """p7_FLogsum: table entry i = float(log(1+exp(-i/1000))) (logsum.c:89), truncating index (logsum.c:110)."""
    L_ = ol.lib()
    tab = np.ctypeslib.as_array(L_.bo_flogsum_table(), shape=(16000,))
    assert tab[0] == np.float32(np.log(2.0)) and tab[15999] > 0
    assert L_.bo_flogsum(1.0, -np.inf) == 1.0
    assert L_.bo_flogsum(0.0, -15.7) == 0.0
    assert L_.bo_flogsum(2.0, 1.9995) == np.float32(2.0) + tab[0]


def test_orf_finder_basic():
    L_ = ol.lib()
    basic = np.zeros(64, np.uint8); L_.bo_gencode_basic(1, ol.u8(basic))
    # 25 x GCT (Ala) then TAA then 19 x GCT: one ORF of 25 aa in frame 0; the 19-aa run is below -l 20
    dna = ol.digitize_dna("GCT" * 25 + "TAA" + "GCT" * 19)
    blk = ol.OrfBlock(); L_.bo_orfblock_init(C.byref(blk))
    L_.bo_translate_orfs(ol.u8(ol.dsq_from(dna)), len(dna), ol.u8(basic), 20, C.byref(blk))
    f0 = [blk.orf[i] for i in range(blk.count) if blk.orf[i].frame == 0]
    assert len(f0) == 1 and (f0[0].start, f0[0].end, f0[0].n) == (1, 75, 25)
    L_.bo_orfblock_free(C.byref(blk))


def test_frameshift_stage_matches_recorded_fs_run():
    """tutorial/AMP_N-fs.out (bathsearch --fs): the only number the reference prints about the frameshift stage is
    'Residues passing Fwd filter', which in this mode is counted inside p7_pli_Frameshift (p7_pipeline.c:1468,1490):
    the length of every DNA window that takes the frameshift branch plus 3n for ORFs aligned by the standard branch.
    411 = the single window (the whole 411-nt target) taking the frameshift branch."""
    assert _printed_counters("AMP_N-fs.out", 0) == (822, 537, 537, 393, 411)
    m = ol.Model(ol.GOLDEN + "/AMP_N.bhmm", 0)
    seqs = [ol.digitize_dna(s) for _, s in ol.read_fasta(ol.GOLDEN + "/target-AMP_N.fa")]
    pli, _, _, fw, _ = m.run_pipeline_fs(seqs)
    assert (pli.nres, pli.pos_past_msv, pli.pos_past_bias, pli.pos_past_vit, pli.pos_past_fwd) == (822, 537, 537, 393, 411)
    assert len(fw) == 1 and (fw[0].strand, fw[0].n, fw[0].length, fw[0].orf_cnt, fw[0].branch) == (0, 1, 411, 3, 1)


def test_frameshift_stage_separates_shifted_from_unshifted_targets():
    """testsuite 2OG-FeII_Oxy_3: the same ten genes with and without frameshifts.  Unshifted genes are one ORF and stay
    on the standard branch; genes whose frameshifts split them into several ORFs go to the frameshift branch."""
    m = ol.Model(ol.GOLDEN + "/2OG-FeII_Oxy_3.bhmm", 0)
    plain = [ol.digitize_dna(s) for _, s in ol.read_fasta(ol.GOLDEN + "/2OG-FeII_Oxy_3-nt.fa")]
    shifted = [ol.digitize_dna(s) for _, s in ol.read_fasta(ol.GOLDEN + "/2OG-FeII_Oxy_3-nt-fs.fa")]
    _, _, _, fw_plain, _ = m.run_pipeline_fs(plain)
    _, _, _, fw_shift, _ = m.run_pipeline_fs(shifted)
    assert len(fw_plain) == 10 and all(w.orf_cnt == 1 and w.branch == 2 for w in fw_plain)
    assert len(fw_shift) == 10
    assert all(w.branch == 1 for w in fw_shift if w.orf_cnt > 1) and sum(w.orf_cnt > 1 for w in fw_shift) >= 3
    for w in fw_plain + fw_shift:                       # every window covers its whole (short) target
        assert w.n == 1 and w.P_fs < 1e-20


def test_frameshift_hit_matches_recorded_run():
    """tutorial/AMP_N-fs.out / .tbl: the one hit of the recorded `bathsearch --fs` run.

        score 82.8  bias 0.1  hmm 1..131  ali 1..402  shifts 6  E-value 1.9e-27

    This goes through every frameshift recursion of the path: 3-codon Forward/Backward parsers, domain decoding, region
    heuristics, 5-codon Forward/Backward, posterior decoding, optimal-accuracy fill and traceback, null2, and the score
    corrections of p7_pli_postDomainDef_Frameshift_BATH.  The E-value is exp(lnP) * Z with the residue count the
    reference reports hits against (822/3 amino-acid positions over max_length 167)."""
    tbl = open(ol.GOLDEN + "/AMP_N-fs.tbl").read().splitlines()[2].split()
    assert tbl[6:8] == ["1", "131"] and tbl[9:11] == ["1", "402"] and tbl[11:14] == ["1.9e-27", "82.8", "0.1"] and tbl[15] == "6"
    m = ol.Model(ol.GOLDEN + "/AMP_N.bhmm", 0)
    seqs = [ol.digitize_dna(s) for _, s in ol.read_fasta(ol.GOLDEN + "/target-AMP_N.fa")]
    pli, fw, _, dm, _, nskip = m.run_pipeline_fsdom(seqs)
    assert len(fw) == 1 and fw[0].branch == 1 and fw[0].ndom == 1 and len(dm) == 1 and nskip == 0
    d = dm[0]
    assert (d.ihmm, d.jhmm, d.iali, d.jali, d.reported, d.n_shifted_codons) == (1, 131, 1, 402, 1, 6)
    assert "%.1f" % d.bitscore == "82.8"
    assert "%.1f" % (d.dombias / np.log(2.0)) == "0.1"
    evalue = np.exp(d.lnP) * (822 / 3.0) / m.om.contents.max_length
    assert "%.1e" % evalue in ("1.9e-27", "1.8e-27", "2.0e-27")


RECORDED_STD_HITS = {   # tutorial/PTH2.tbl and tutorial/AMP_N.out: (hmm from, hmm to, ali from, ali to, score, bias)
    "PTH2.bhmm": ("target-PTH2.fa", [(2, 116, 672, 325, "110.6", "0.3"), (35, 116, 1486, 1731, "86.4", "0.0"),
                                     (71, 113, 2468, 2343, "36.2", "0.0"), (2, 30, 1273, 1359, "36.0", "0.3")]),
    "AMP_N.bhmm": ("target-AMP_N.fa", [(None, None, 7, 234, "47.8", "0.0")]),
}


@pytest.mark.parametrize("hmmfile", sorted(RECORDED_STD_HITS))
def test_standard_branch_hits_match_recorded_runs(hmmfile):
    """The hits of the recorded plain `bathsearch` runs: every hit of tutorial/PTH2.tbl (both strands) and the hit of
    tutorial/AMP_N.out, to the printed digits.  Path: cascade -> Forward/Backward parsers -> domain decoding -> region
    heuristics -> full Forward/Backward of the envelope -> posterior decoding -> optimal accuracy fill and traceback ->
    null2 -> p7_pli_postDomainDef_BATH's score arithmetic and coordinate mapping."""
    fasta, want = RECORDED_STD_HITS[hmmfile]
    if hmmfile == "PTH2.bhmm":                       # the fixture really holds these rows
        rows = [l.split() for l in open(ol.GOLDEN + "/PTH2.tbl") if l and l[0] != "#"]
        assert [(int(r[6]), int(r[7]), int(r[9]), int(r[10]), r[12], r[13]) for r in rows] == want
    m = ol.Model(ol.GOLDEN + "/" + hmmfile, 0)
    seqs = [ol.digitize_dna(s) for _, s in ol.read_fasta(ol.GOLDEN + "/" + fasta)]
    pli, dm, _, nskip = m.run_pipeline_hits(seqs)
    got = sorted(dm, key=lambda d: -d.bitscore)
    assert nskip == 0 and len(got) == len(want) and all(d.reported for d in got)
    for d, (h1, h2, a1, a2, score, bias) in zip(got, want):
        assert (d.iali, d.jali) == (a1, a2)
        if h1 is not None:
            assert (d.ihmm, d.jhmm) == (h1, h2)
        assert "%.1f" % d.bitscore == score and "%.1f" % (d.dombias / np.log(2.0)) == bias


def recorded_envelopes():
    """Envelope coordinates the reference recorded: tutorial/PTH2-cigar.tbl (an earlier --tblout layout that still printed
    'env from / env to') for the four PTH2 hits, and the hit line of tutorial/AMP_N-frameline.out (bathsearch --fs
    --frameline prints env-from / env-to) for the frameshift hit."""
    rows = [l.split() for l in open(ol.GOLDEN + "/PTH2-cigar.tbl") if l and l[0] != "#"]
    pth2 = [(int(r[9]), int(r[10]), int(r[11]), int(r[12]), r[14], r[17]) for r in rows]      # ali from/to, env from/to, score, CIGAR
    fl = [l.split() for l in open(ol.GOLDEN + "/AMP_N-frameline.out") if l.startswith(" ! ")][0]
    amp = (int(fl[7]), int(fl[8]), int(fl[10]), int(fl[11]), fl[1])                           # ali 1..402, env 1..411, score 82.8
    return pth2, amp


def test_envelope_coordinates_match_recorded_runs():
    """Pins the region heuristics' end points (p7_domaindef.c:355-372 / :546-560), not only the alignment's."""
    pth2, amp = recorded_envelopes()
    assert pth2 == [(672, 325, 675, 325, "110.6", "99M6I192M3D51M"), (1486, 1731, 1444, 1731, "86.4", "246M"),
                    (2468, 2343, 2483, 2325, "36.2", "42M3D84M"), (1273, 1359, 1270, 1383, "36.0", "87M")]
    assert amp == (1, 402, 1, 411, "82.8")
    m = ol.Model(ol.GOLDEN + "/PTH2.bhmm", 0)
    seqs = [ol.digitize_dna(s) for _, s in ol.read_fasta(ol.GOLDEN + "/target-PTH2.fa")]
    _, dm, _, _ = m.run_pipeline_hits(seqs)
    got = sorted(dm, key=lambda d: -d.bitscore)
    assert [(d.iali, d.jali, d.ienv, d.jenv, "%.1f" % d.bitscore) for d in got] == [t[:5] for t in pth2]
    m = ol.Model(ol.GOLDEN + "/AMP_N.bhmm", 0)
    seqs = [ol.digitize_dna(s) for _, s in ol.read_fasta(ol.GOLDEN + "/target-AMP_N.fa")]
    _, _, _, dm, _, _ = m.run_pipeline_fsdom(seqs)
    assert [(d.iali, d.jali, d.ienv, d.jenv, "%.1f" % d.bitscore) for d in dm] == [amp]
