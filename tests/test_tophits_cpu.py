"""The hit-list stage (bath_tophits_*: host code of libbathhip) without a GPU: fed with the ORACLE's domains of the recorded
runs it must reproduce every column of tutorial/PTH2.tbl and tutorial/AMP_N-fs.tbl that does not come from the alignment
display (percent identity, stops, CIGAR are GPU-path outputs, checked in tests/test_tblout_gpu.py); plus the duplicate
removal and ordering rules of p7_tophits.c on constructed hits."""
import ctypes as C

import numpy as np
import pytest

import bath_amd as ba
import oracle_lib as ol


def from_oracle(o, window):
    d = ba.FsDomain()
    for f in ("ienv", "jenv", "iali", "jali", "ihmm", "jhmm", "envsc", "oasc", "domcorrection", "dombias", "bitscore", "pre_score", "lnP",
              "reported", "n_shifted_codons"):
        setattr(d, f, getattr(o, f))
    d.window = window
    d.strand = 1 if o.iali > o.jali else 0
    return d


def data_rows(text):
    return [l.split() for l in text.split("\n") if l and l[0] != "#"]


@pytest.mark.parametrize("hmmfile,fasta,fs,golden", [("PTH2.bhmm", "target-PTH2.fa", False, "PTH2.tbl"),
                                                      ("AMP_N.bhmm", "target-AMP_N.fa", True, "AMP_N-fs.tbl")])
def test_table_columns_from_oracle_domains(hmmfile, fasta, fs, golden):
    m = ol.Model(ol.GOLDEN + "/" + hmmfile, 0)
    hmm = ba.HMM(ol.GOLDEN + "/" + hmmfile, 0)
    recs = ol.read_fasta(ol.GOLDEN + "/" + fasta)
    seqs = [ol.digitize_dna(s) for _, s in recs]
    if fs:
        pli, _, _, odm, per_d, _ = m.run_pipeline_fsdom(seqs)
    else:
        pli, odm, per_d, _ = m.run_pipeline_hits(seqs)
    doms = [from_oracle(o, w) for w, (a, b) in enumerate(per_d) for o in odm[a:b]]
    th = ba.TopHits()
    th.add(doms, [n.split()[0] for n, _ in recs], [len(s) for s in seqs])
    th.finalize(pli.nres, hmm.max_length)
    text = th.tblout(hmm.name, hmm.acc, hmm.M, fs_pipe=fs, show_cigar=True)
    want_lines = open(ol.GOLDEN + "/" + golden).read().split("\n")
    assert text.split("\n")[:2] == want_lines[:2]                    # both header lines, byte for byte
    got, want = data_rows(text), data_rows("\n".join(want_lines))
    assert len(got) == len(want)
    ncmp = 14                                                         # hit ID .. bias; then PID [shifts stops] CIGAR
    for g, w in zip(got, want):
        assert g[:ncmp] == w[:ncmp]
        if fs:
            assert g[15] == w[15]                                     # shifts


def mk(window, iali, jali, ihmm, jhmm, lnP, score=50.0):
    d = ba.FsDomain()
    d.window, d.iali, d.jali, d.ienv, d.jenv, d.ihmm, d.jhmm, d.lnP, d.bitscore, d.reported = window, iali, jali, iali, jali, ihmm, jhmm, lnP, score, 1
    d.strand = 1 if iali > jali else 0
    return d


def test_duplicates_sorting_and_threshold():
    doms = [mk(0, 100, 400, 1, 100, -40.0),          # kept
            mk(0, 102, 380, 5, 90, -30.0),           # same strand, flush start, overlapping model range: duplicate of the first
            mk(0, 400, 100, 1, 100, -35.0),          # other strand: kept
            mk(0, 1000, 1300, 1, 100, -20.0),        # elsewhere: kept
            mk(1, 100, 400, 1, 100, -50.0),          # other sequence: kept, best E-value
            mk(0, 5000, 5300, 1, 100, -1.0),         # E-value above the threshold after the search-space correction
            mk(0, 7000, 7300, 1, 100, -45.0)]        # not reported by the pipeline: never becomes a hit
    doms[-1].reported = 0
    th = ba.TopHits()
    th.add(doms, ["a", "b"], [10000, 10000])
    th.finalize(nres=3000 * 100, max_length=100)     # log(nres / (3 * max_length)) = log(1000)
    hits = th.hits()
    assert len(hits) == 6
    by_key = {(idx, d.iali): (d, fl) for d, idx, fl in hits}
    assert [idx for _, idx, _ in hits][0] == 1                                            # sorted by E-value
    lnps = [d.lnP for d, _, _ in hits]
    assert lnps == sorted(lnps)
    assert abs(by_key[(0, 100)][0].lnP - (-40.0 + np.log(np.float32(300000.0) / np.float32(300.0)))) < 1e-9
    assert by_key[(0, 102)][1] & 4 and not by_key[(0, 102)][1] & 1                        # duplicate, not reported
    for key in ((0, 100), (0, 400), (0, 1000), (1, 100)):
        assert by_key[key][1] == 3                                                        # reported and included (E <= 0.01)
    assert by_key[(0, 5000)][1] == 0                                                      # exp(-1) * 1000 > 10
    text = th.tblout("q", "", 100)
    rows = data_rows(text)
    assert [r[0] for r in rows] == ["1", "2", "3", "4"] and rows[0][1] == "b" and all(r[4] == "-" for r in rows)


def test_empty_list_prints_header_only():
    th = ba.TopHits()
    th.finalize(1000, 100)
    text = th.tblout("query", "ACC1", 50, fs_pipe=True, show_cigar=False)
    lines = text.split("\n")
    assert len(lines) == 3 and lines[2] == "" and lines[0].startswith("# hit ID") and "shifts" in lines[0] and lines[0].endswith("description of target")


import recorded


@pytest.mark.parametrize("outfile,hmmfile,fasta,fs", recorded.RUNS, ids=[r[0] for r in recorded.RUNS])
def test_oracle_reproduces_recorded_annotation_lines(outfile, hmmfile, fasta, fs):
    """All 12 per-hit lines of tutorial/*.out from the ORACLE's domains (stop counts excepted: an alignment-display output
    the oracle does not restate).  Pins oracle/domaindef.c and oracle/fs_domaindef.c on two more models and codon table 4."""
    want = recorded.annotation_lines(outfile)
    recs = ol.read_fasta(ol.GOLDEN + "/" + fasta)
    seqs = [ol.digitize_dna(s) for _, s in recs]
    for q, lines in enumerate(want):
        m = ol.Model(ol.GOLDEN + "/" + hmmfile, q)
        hmm = ba.HMM(ol.GOLDEN + "/" + hmmfile, q)
        if fs:
            pli, _, _, odm, per_d, _ = m.run_pipeline_fsdom(seqs)
        else:
            pli, odm, per_d, _ = m.run_pipeline_hits(seqs)
        doms = [from_oracle(o, w) for w, (a, b) in enumerate(per_d) for o in odm[a:b]]
        th = ba.TopHits()
        th.add(doms, [n.split()[0] for n, _ in recs], [len(s) for s in seqs])
        th.finalize(pli.nres, hmm.max_length)
        got = [recorded.fields_of(d, len(seqs[idx]), fs, with_env=len(lines[0]) == 11) for d, idx, fl in th.hits() if fl & 1]
        if fs:
            for g, w in zip(got, lines):
                g[8] = w[8]                                            # stops
        assert got == lines


@pytest.mark.parametrize("outfile,hmmfile,fasta,fs", recorded.RUNS, ids=[r[0] for r in recorded.RUNS])
def test_targets_block_matches_recorded_output(outfile, hmmfile, fasta, fs):
    """The 'Scores for complete hits' block of the reference's main output, byte for byte, from the oracle's domains (the stop
    count of the --fs hit is an alignment-display output and is filled in from the record)."""
    want = recorded.targets_blocks(outfile)
    recs = ol.read_fasta(ol.GOLDEN + "/" + fasta)
    seqs = [ol.digitize_dna(s) for _, s in recs]
    for q, block in enumerate(want):
        m = ol.Model(ol.GOLDEN + "/" + hmmfile, q)
        hmm = ba.HMM(ol.GOLDEN + "/" + hmmfile, q)
        if fs:
            pli, _, _, odm, per_d, _ = m.run_pipeline_fsdom(seqs)
        else:
            pli, odm, per_d, _ = m.run_pipeline_hits(seqs)
        doms = [from_oracle(o, w) for w, (a, b) in enumerate(per_d) for o in odm[a:b]]
        if fs:
            for d in doms:
                d.n_stops = 1                                      # tutorial/AMP_N-fs.out: 1 stop codon in the alignment
        th = ba.TopHits()
        th.add(doms, [n.split()[0] for n, _ in recs], [len(s) for s in seqs])
        th.finalize(pli.nres, hmm.max_length)
        assert th.targets(fs_pipe=fs) == block


@pytest.mark.parametrize("outfile,hmmfile,fasta,fs", [("PTH2.out", "PTH2.bhmm", "target-PTH2.fa", False), ("AMP_N-fs.out", "AMP_N.bhmm", "target-AMP_N.fa", True)])
def test_annotation_heads_match_recorded_output(outfile, hmmfile, fasta, fs):
    """'>> seq1', header lines and the hit line of each entry under 'Annotation for each hit', byte for byte, for the two
    recorded outputs written by the reference version in /root/reference (the other two predate a column change)."""
    m = ol.Model(ol.GOLDEN + "/" + hmmfile, 0)
    hmm = ba.HMM(ol.GOLDEN + "/" + hmmfile, 0)
    recs = ol.read_fasta(ol.GOLDEN + "/" + fasta)
    seqs = [ol.digitize_dna(s) for _, s in recs]
    if fs:
        pli, _, _, odm, per_d, _ = m.run_pipeline_fsdom(seqs)
    else:
        pli, odm, per_d, _ = m.run_pipeline_hits(seqs)
    doms = [from_oracle(o, w) for w, (a, b) in enumerate(per_d) for o in odm[a:b]]
    for d in doms:
        d.n_stops = 1 if fs else 0
    th = ba.TopHits()
    th.add(doms, [n.split()[0] for n, _ in recs], [len(s) for s in seqs])
    th.finalize(pli.nres, hmm.max_length)
    assert th.annotations(hmm.M, fs_pipe=fs) == recorded.annotation_heads(outfile)


@pytest.mark.parametrize("outfile,hmmfile,fasta,fs", recorded.RUNS, ids=[r[0] for r in recorded.RUNS])
def test_statistics_block_matches_recorded_output(outfile, hmmfile, fasta, fs):
    """The 'Internal pipeline statistics summary' block (counters, fractions, hit count and covered fraction), byte for byte."""
    want = recorded.statistics_blocks(outfile)
    recs = ol.read_fasta(ol.GOLDEN + "/" + fasta)
    seqs = [ol.digitize_dna(s) for _, s in recs]
    for q, block in enumerate(want):
        m = ol.Model(ol.GOLDEN + "/" + hmmfile, q)
        hmm = ba.HMM(ol.GOLDEN + "/" + hmmfile, q)
        if fs:
            pli, _, _, odm, per_d, _ = m.run_pipeline_fsdom(seqs)
        else:
            pli, odm, per_d, _ = m.run_pipeline_hits(seqs)
        doms = [from_oracle(o, w) for w, (a, b) in enumerate(per_d) for o in odm[a:b]]
        th = ba.TopHits()
        th.add(doms, [n.split()[0] for n, _ in recs], [len(s) for s in seqs])
        th.finalize(pli.nres, hmm.max_length)
        st = ba.PipelineStats()
        for f in ("nres", "pos_past_msv", "pos_past_bias", "pos_past_vit", "pos_past_fwd"):
            setattr(st, f, getattr(pli, f))
        prm = ba.PipelineParams()
        ba.lib().bath_pipeline_params_default(C.byref(prm), 1 if fs else 0)
        assert th.statistics(st, prm, 1, hmm.M, len(seqs)) == block


def test_add_arrays_gives_the_same_table_as_add():
    """TopHits.add_arrays (one record array + one CIGAR pool, as Pipeline.run_hits(arrays=True) and dist.gather_query_hits deliver
    them) against add() on the same hits, including a HitArray that went through to_bytes / from_bytes / concat."""
    import bath_amd as ba
    doms = []
    for i in range(7):
        d = ba.FsDomain()
        d.window, d.iali, d.jali, d.ienv, d.jenv, d.ihmm, d.jhmm = i % 3, 100 + 40 * i, 400 + 40 * i, 90 + 40 * i, 410 + 40 * i, 1, 90
        d.bitscore, d.lnP, d.reported, d.pre_score = 40.0 + 3 * i, -25.0 - 2 * i, 1, 41.0 + 3 * i
        d.cigar = "%dM2I%dM" % (10 + i, 20 + i)
        doms.append(d)
    names, lens = ["a", "b", "c"], [5000, 6000, 7000]

    def table(fill):
        th = ba.TopHits()
        fill(th)
        th.finalize(3_000_000, 100)
        return th.tblout("q", "", 90, show_cigar=True)

    want = table(lambda th: th.add(doms, names, lens))
    whole = ba.HitArray.from_domains(doms)
    assert table(lambda th: th.add_arrays(whole, names, lens)) == want
    a, b = ba.HitArray.from_domains(doms[:3]), ba.HitArray.from_domains(doms[3:])
    buf = a.to_bytes() + b.to_bytes()
    a2, p = ba.HitArray.from_bytes(buf, 0)
    b2, p = ba.HitArray.from_bytes(buf, p)
    assert p == len(buf)
    assert table(lambda th: th.add_arrays(ba.HitArray.concat([a2, b2]), names, lens)) == want
