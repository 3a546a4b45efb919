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
