"""Every per-hit annotation line the reference recorded (tutorial/PTH2.out, AMP_N.out, MET-ct4.out [two models, codon
table 4], AMP_N-fs.out: 12 hits) from the GPU path: score, bias, E-value, model / alignment / envelope coordinates,
frameshift and stop counts, sequence length and accuracy, to the printed digits."""
import pytest

import bath_amd as ba
import oracle_lib as ol
import recorded

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("outfile,hmmfile,fasta,fs", recorded.RUNS, ids=[r[0] for r in recorded.RUNS])
def test_annotation_lines(outfile, hmmfile, fasta, fs):
    ctx = ba.Context(0)
    want = recorded.annotation_lines(outfile)
    recs = ol.read_fasta(ol.GOLDEN + "/" + fasta)
    seqs = [ba.digitize(s, ba.DNA_SYMS) for _, s in recs]
    assert len(want) == ba.HMM.count(ol.GOLDEN + "/" + hmmfile)
    for q, lines in enumerate(want):
        hmm = ba.HMM(ol.GOLDEN + "/" + hmmfile, q)
        om = ba.OProfile(ctx, ba.Profile(hmm))
        pipe = ba.Pipeline(ctx, om, fs_pipe=fs, ncbi_table=hmm.ct)
        block = ba.SeqBlock(ctx, seqs)
        if fs:
            om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
            om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
            stats, _, dm, nskip = pipe.run_frameshift_domains(om3, om5, block)
        else:
            stats, dm, nskip = pipe.run_hits(block)
        th = ba.TopHits()
        th.add(dm, [n.split()[0] for n, _ in recs], [len(s) for s in seqs])
        th.finalize(stats.nres, hmm.max_length)
        got = [recorded.fields_of(d, len(seqs[idx]), fs, with_env=len(lines[0]) == 11) for d, idx, fl in th.hits() if fl & 1]
        assert got == lines


def test_envelope_coordinates_match_recorded_runs():
    """tutorial/PTH2-cigar.tbl (earlier --tblout layout with 'env from / env to') and the hit line of
    tutorial/AMP_N-frameline.out: envelope end points of all five recorded hits, from the GPU path."""
    from test_oracle_cpu import recorded_envelopes
    pth2, amp = recorded_envelopes()
    ctx = ba.Context(0)
    for hmmfile, fasta, fs, want in (("PTH2.bhmm", "target-PTH2.fa", False, [t[:5] for t in pth2]), ("AMP_N.bhmm", "target-AMP_N.fa", True, [amp])):
        hmm = ba.HMM(ol.GOLDEN + "/" + hmmfile, 0)
        seqs = [ba.digitize(s, ba.DNA_SYMS) for _, s in ol.read_fasta(ol.GOLDEN + "/" + fasta)]
        om = ba.OProfile(ctx, ba.Profile(hmm))
        pipe = ba.Pipeline(ctx, om, fs_pipe=fs, ncbi_table=hmm.ct)
        block = ba.SeqBlock(ctx, seqs)
        if fs:
            om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
            om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
            _, _, dm, _ = pipe.run_frameshift_domains(om3, om5, block)
        else:
            _, dm, _ = pipe.run_hits(block)
        got = sorted(dm, key=lambda d: -d.bitscore)
        assert [(d.iali, d.jali, d.ienv, d.jenv, "%.1f" % d.bitscore) for d in got] == want
    if True:
        rows = [d.cigar for d in got]
        assert rows == ["44M1F39M1B114M9I25M2B19M1B44M1B4M6I30M2B67M"]          # AMP_N-fs.tbl's CIGAR: the frameline alignment's columns
