"""End to end against the reference's own recorded tables: model file + target FASTA -> GPU pipeline -> hit list ->
--tblout text, compared byte for byte with tutorial/PTH2.tbl (bathsearch --cigar) and tutorial/AMP_N-fs.tbl
(bathsearch --fs --cigar), every column: coordinates, E-value, score, bias, percent identity, frameshift and stop counts,
CIGAR.  Only the trailer (program name, paths, date) is not reproduced."""
import pytest

import bath_amd as ba
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def table_body(path):
    lines = open(path).read().split("\n")
    cut = lines.index("#")                       # the trailer starts with a bare '#'
    return "\n".join(lines[:cut]) + "\n"


def search(ctx, hmmfile, fasta, fs):
    hmm = ba.HMM(ol.GOLDEN + "/" + hmmfile, 0)
    recs = ol.read_fasta(ol.GOLDEN + "/" + fasta)
    names = [n.split()[0] for n, _ in recs]
    seqs = [ba.digitize(s, ba.DNA_SYMS) for _, s in recs]
    om = ba.OProfile(ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(ctx, om, fs_pipe=fs, ncbi_table=hmm.ct)
    block = ba.SeqBlock(ctx, seqs)
    if fs:
        om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
        om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
        stats, _, dm, _ = pipe.run_frameshift_domains(om3, om5, block)
    else:
        stats, dm, _ = pipe.run_hits(block)
    th = ba.TopHits()
    th.add(dm, names, [len(s) for s in seqs])
    th.finalize(stats.nres, hmm.max_length)
    return th.tblout(hmm.name, hmm.acc, hmm.M, fs_pipe=fs, show_cigar=True)


@pytest.mark.parametrize("hmmfile,fasta,fs,golden", [("PTH2.bhmm", "target-PTH2.fa", False, "PTH2.tbl"),
                                                      ("AMP_N.bhmm", "target-AMP_N.fa", True, "AMP_N-fs.tbl")])
def test_tblout_matches_recorded_table(hmmfile, fasta, fs, golden):
    ctx = ba.Context(0)
    assert search(ctx, hmmfile, fasta, fs) == table_body(ol.GOLDEN + "/" + golden)


@pytest.mark.parametrize("outfile,hmmfile,fasta,fs", [("PTH2.out", "PTH2.bhmm", "target-PTH2.fa", False), ("AMP_N-fs.out", "AMP_N.bhmm", "target-AMP_N.fa", True),
                                                      ("AMP_N.out", "AMP_N.bhmm", "target-AMP_N.fa", False)])
def test_targets_block_matches_recorded_output(outfile, hmmfile, fasta, fs):
    """'Scores for complete hits' of the reference's main output, byte for byte, from the GPU path."""
    import recorded
    ctx = ba.Context(0)
    hmm = ba.HMM(ol.GOLDEN + "/" + hmmfile, 0)
    recs = ol.read_fasta(ol.GOLDEN + "/" + fasta)
    seqs = [ba.digitize(s, ba.DNA_SYMS) for _, s in recs]
    om = ba.OProfile(ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(ctx, om, fs_pipe=fs, ncbi_table=hmm.ct)
    block = ba.SeqBlock(ctx, seqs)
    if fs:
        om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
        om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
        stats, _, dm, _ = pipe.run_frameshift_domains(om3, om5, block)
    else:
        stats, dm, _ = pipe.run_hits(block)
    th = ba.TopHits()
    th.add(dm, [n.split()[0] for n, _ in recs], [len(s) for s in seqs])
    th.finalize(stats.nres, hmm.max_length)
    assert th.targets(fs_pipe=fs) == recorded.targets_blocks(outfile)[0]


@pytest.mark.parametrize("outfile,hmmfile,fasta,fs", [("PTH2.out", "PTH2.bhmm", "target-PTH2.fa", False), ("AMP_N-fs.out", "AMP_N.bhmm", "target-AMP_N.fa", True)])
def test_annotation_heads_match_recorded_output(outfile, hmmfile, fasta, fs):
    """'>> name', header lines and hit line of every entry under 'Annotation for each hit', byte for byte, from the GPU path."""
    import recorded
    ctx = ba.Context(0)
    hmm = ba.HMM(ol.GOLDEN + "/" + hmmfile, 0)
    recs = ol.read_fasta(ol.GOLDEN + "/" + fasta)
    seqs = [ba.digitize(s, ba.DNA_SYMS) for _, s in recs]
    om = ba.OProfile(ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(ctx, om, fs_pipe=fs, ncbi_table=hmm.ct)
    block = ba.SeqBlock(ctx, seqs)
    if fs:
        om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
        om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
        stats, _, dm, _ = pipe.run_frameshift_domains(om3, om5, block)
    else:
        stats, dm, _ = pipe.run_hits(block)
    th = ba.TopHits()
    th.add(dm, [n.split()[0] for n, _ in recs], [len(s) for s in seqs])
    th.finalize(stats.nres, hmm.max_length)
    assert th.annotations(hmm.M, fs_pipe=fs) == recorded.annotation_heads(outfile)
    # ... and the 'Internal pipeline statistics summary' block from the GPU path's counters
    assert th.statistics(stats, pipe.params, 1, hmm.M, len(seqs)) == recorded.statistics_blocks(outfile)[0]
