"""Every alignment block of the reference's recorded runs (tutorial/*.out: per hit the model row, the match row, the
translation row, the codon row with its frameshift marks, the optional frame row and a posterior-probability digit per column)
against what the ORACLE's traces give -- without a GPU.

This pins rows a8 / a9 column by column instead of through one score printed to a decimal: 131 columns with six shifted codons
and a stop for the frameshift hit (AMP_N-fs.out, and AMP_N-frameline.out with the frame row), 4 + 1 + 6 hits of the standard
branch (PTH2.out with its CS line, AMP_N.out, MET-ct4.out under codon table 4).  The text is produced by the product's host-side
renderer (bath_alidisplay_print: p7_alidisplay_fs_Create / _nonfs_Create + p7_alidisplay_Print_BATH) from the oracle's dom->tr;
tests/test_alidisplay_gpu.py feeds the same renderer the GPU path's traces."""
import numpy as np
import pytest

import bath_amd as ba
import oracle_lib as ol

RUNS = [  # output file, model file, target FASTA, --fs, --frameline
    ("PTH2.out", "PTH2.bhmm", "target-PTH2.fa", False, False),
    ("AMP_N.out", "AMP_N.bhmm", "target-AMP_N.fa", False, False),
    ("MET-ct4.out", "MET-ct4.bhmm", "target-MET.fa", False, False),
    ("AMP_N-fs.out", "AMP_N.bhmm", "target-AMP_N.fa", True, False),
    ("AMP_N-frameline.out", "AMP_N.bhmm", "target-AMP_N.fa", True, True),
]


def recorded_blocks(outfile):
    """[[(ali_from, ali_to, block text) per hit] per query]: the lines between 'score: ... bits' and the hit's last PP line."""
    lines = open(ol.GOLDEN + "/" + outfile).read().split("\n")
    per_query, cur, i = [], None, 0
    while i < len(lines):
        if lines[i].startswith("Query:"):
            cur = []
            per_query.append(cur)
        if lines[i].startswith(" ! ") or lines[i].startswith(" ? "):
            hit_line = lines[i]
        if lines[i].startswith("  score: ") and lines[i].endswith(" bits"):
            j = i + 1
            body = []
            while j < len(lines) and not lines[j].startswith(">>") and not lines[j].startswith("Internal pipeline"):
                body.append(lines[j])
                j += 1
            while body and not body[-1].strip():
                body.pop()
            toks = [t for t in hit_line.split()[1:] if not set(t) <= set("[].")]
            cur.append((int(toks[5]), int(toks[6]), "\n".join(body) + "\n"))
            i = j - 1
        i += 1
    return per_query


def oracle_hits(model, seqs, fs):
    ol.lib().bo_traces_reset()
    if fs:
        pli, _, _, odm, per_d, _ = model.run_pipeline_fsdom(seqs)
    else:
        pli, odm, per_d, _ = model.run_pipeline_hits(seqs)
    return [(w, o) for w, (a, b) in enumerate(per_d) for o in odm[a:b] if o.reported]


def strand_codes(codes, bottom):
    return (np.where(codes < 4, 3 - codes, np.array([ol.lib().bo_dna_complement(int(x)) for x in codes], dtype=np.uint8))[::-1] if bottom else codes).astype(np.uint8)


def render(hmm, gm, gm5, trace, codes_by_strand, d, seq_name, frameline):
    t = trace[0]
    bottom = d.iali > d.jali
    window = codes_by_strand[1 if bottom else 0][t.win_start - 1:]
    return ba.alidisplay_print(trace, window, hmm, d.iali, d.jali, seq_name, gm_fs5=gm5, gm=gm, ncbi_table=hmm.ct, frameline=frameline)


@pytest.mark.parametrize("outfile,hmmfile,fasta,fs,frameline", RUNS)
def test_oracle_traces_reproduce_recorded_alignment_blocks(outfile, hmmfile, fasta, fs, frameline):
    want = recorded_blocks(outfile)
    recs = ol.read_fasta(ol.GOLDEN + "/" + fasta)
    seqs = [ol.digitize_dna(s) for _, s in recs]
    assert len(want) == ba.HMM.count(ol.GOLDEN + "/" + hmmfile)
    ncols = 0
    for q, blocks in enumerate(want):
        model = ol.Model(ol.GOLDEN + "/" + hmmfile, q)
        hmm = ba.HMM(ol.GOLDEN + "/" + hmmfile, q)
        gm, gm5 = ba.Profile(hmm), ba.FSProfile(hmm, 5, ncbi_table=hmm.ct)
        hits = oracle_hits(model, seqs, fs)
        assert len(hits) == len(blocks)
        by_ali = {(o.iali, o.jali): (w, o) for w, o in hits}
        for a, b, text in blocks:
            w, o = by_ali[(a, b)]
            trace = ol.trace_arrays(o.trace_idx)
            assert trace[0].frameshift == (1 if fs else 0)
            both = [seqs[w], strand_codes(seqs[w], True)]
            got = render(hmm, gm, gm5, trace, both, o, recs[w][0].split()[0], frameline)
            assert got == text, "\n" + got + "\n--- recorded ---\n" + text
            ncols += trace[0].N
    assert ncols >= 60


def test_renderer_line_width():
    """--textw: a block holds (textw - names - 2 x coordinates - 9) / 5 columns.  --notextw (textw <= 0) is the reference's own
    arithmetic too: max_aliwidth = ad->N, then the same "- 4, / 5" (p7_alidisplay.c:3806-3810), so (N - 4) / 5 columns per block."""
    model = ol.Model(ol.GOLDEN + "/AMP_N.bhmm", 0)
    hmm = ba.HMM(ol.GOLDEN + "/AMP_N.bhmm", 0)
    recs = ol.read_fasta(ol.GOLDEN + "/target-AMP_N.fa")
    seqs = [ol.digitize_dna(s) for _, s in recs]
    (w, o), = oracle_hits(model, seqs, True)
    trace = ol.trace_arrays(o.trace_idx)
    N = trace[0].N
    gm5 = ba.FSProfile(hmm, 5, ncbi_table=hmm.ct)
    window = seqs[w][trace[0].win_start - 1:]
    blocks = lambda per_line: -(-N // per_line)
    narrow = ba.alidisplay_print(trace, window, hmm, o.iali, o.jali, "seq1", gm_fs5=gm5, textw=120)
    assert narrow.count(" PP\n") == blocks((120 - 8 - 2 * 3 - 5 - 4) // 5)
    unlimited = ba.alidisplay_print(trace, window, hmm, o.iali, o.jali, "seq1", gm_fs5=gm5, textw=0)
    assert unlimited.count(" PP\n") == blocks((N - 4) // 5)
    # whatever the width, the same columns in the same order
    cols = lambda text, tag: "".join("".join(l[:-len(tag)].split()) for l in text.split("\n") if l.endswith(tag))
    default = ba.alidisplay_print(trace, window, hmm, o.iali, o.jali, "seq1", gm_fs5=gm5)
    assert cols(narrow, " PP") == cols(default, " PP") == cols(unlimited, " PP")
