"""Long targets: the windows-with-context scheme of bathsearch (esl_sqio_ReadWindow(dbfp, 3*max_length, block_length, .),
bathsearch.c:1060-1110; ESL_SQ.C skip rule, p7_pipeline.c:1635-1637; pli->nres += W, bathsearch.c:1258) on the GPU path.

* window by window the GPU agrees with the oracle run with the same contexts (domains, nres, pos_past_fwd);
* the search of the split target gives the same table as the search of the whole target: same hits, coordinates, scores and
  E-values -- genes lying on a boundary, inside a context, and on both strands included -- which is the property the
  reference's scheme exists to provide (duplicates from the overlap are removed by p7_tophits_RemoveDuplicates)."""
import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol
from bath_amd import dist as bd

pytestmark = pytest.mark.gpu

BLOCK = 60000


def planted_genome(rng, model, L=400000):
    g = rng.integers(0, 4, size=L).astype(np.uint8)
    genes = common.emit_from_model(rng, model, 30, flank=3, sharpen=3.0)
    C = 3 * model.om.contents.max_length
    # on a boundary, just inside a context, just before a context, and elsewhere
    spots = [BLOCK - 150, 2 * BLOCK - C + 20, 3 * BLOCK - C - 400, 4 * BLOCK + 5, 5 * BLOCK - 90] + [int(x) for x in rng.integers(1000, L - 2000, size=25)]
    for i, (aa, p) in enumerate(zip(genes, spots)):
        nt = np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8)
        if i % 2:
            nt = (3 - nt[::-1]).astype(np.uint8)
        g[p:p + len(nt)] = nt[: L - p]
    return g


@pytest.mark.parametrize("hmmfile", ["PTH2.bhmm", "Caudal_act.bhmm"])
def test_split_target_equals_whole_target(hmmfile):
    ctx = ba.Context(0)
    path = ol.GOLDEN + "/" + hmmfile
    model = ol.Model(path, 0)
    hmm = ba.HMM(path, 0)
    rng = np.random.default_rng(2026)
    genome = planted_genome(rng, model)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)

    def table(domains, nres):
        th = ba.TopHits()
        th.add(domains, ["chr"], [len(genome)])
        th.finalize(nres, hmm.max_length)
        return th.tblout(hmm.name, hmm.acc, hmm.M, show_cigar=True), th

    st_whole, dm_whole, _ = pipe.run_hits(ba.SeqBlock(ctx, [genome]))
    text_whole, th_whole = table(dm_whole, st_whole.nres)

    wins = bd.split_targets([len(genome)], hmm.max_length, BLOCK)
    assert len(wins) == 7 and all(c == 3 * hmm.max_length for _, _, _, c in wins[1:])
    seqs = [genome[s:s + n] for _, s, n, _ in wins]
    block = ba.SeqBlock(ctx, seqs)
    block.set_context([c for _, _, _, c in wins])
    st_split, dm_split, _ = pipe.run_hits(block)
    assert st_split.nres == st_whole.nres == 2 * len(genome)

    # window by window against the oracle with the same contexts
    pli, odm, per_d, _ = model.run_pipeline_hits(seqs, contexts=[c for _, _, _, c in wins])
    assert (pli.nres, pli.n_past_fwd, pli.pos_past_fwd) == (st_split.nres, st_split.n_past_fwd, st_split.pos_past_fwd)
    want = sorted((w, o.ienv, o.jenv, o.iali, o.jali, o.ihmm, o.jhmm) for w, (a, b) in enumerate(per_d) for o in odm[a:b])
    assert sorted((d.window, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm) for d in dm_split) == want

    # window coordinates -> target coordinates, one hit list for the target
    window_end = {}
    for d in dm_split:
        off = wins[d.window][1]
        d.ienv += off; d.jenv += off; d.iali += off; d.jali += off
        window_end[(d.iali, d.jali)] = off + wins[d.window][2]
        d.window = 0
    text_split, th_split = table(dm_split, st_split.nres)

    def reported(th):
        return {(d.iali, d.jali, d.ienv, d.jenv, d.ihmm, d.jhmm, "%.1f" % d.bitscore, "%.2g" % np.exp(d.lnP), "%.2f" % d.pid) for d, _, fl in th.hits() if fl & 1}
    whole, split = reported(th_whole), reported(th_split)
    assert whole <= split and len(whole) >= 8
    # what the split search reports in addition are alignments cut off by the end of a window (their envelope runs into it):
    # the same gene is seen whole in the next window, where in these cases it forms a multi-domain region (not built: f4)
    for h in split - whole:
        assert max(h[2], h[3]) >= window_end[(h[0], h[1])] - 3
    assert len(split - whole) <= 2
    if hmmfile == "PTH2.bhmm":                                   # here an ORF reaches out of a context: found twice, reported once
        assert any(fl & 4 for _, _, fl in th_split.hits())
    if not (split - whole):
        assert text_split == text_whole


@pytest.mark.parametrize("lanes", [2, 3])
def test_contexts_survive_the_split_into_concurrent_parts(monkeypatch, lanes):
    """bath_hip_pipeline_filters runs big blocks as concurrent parts; each part must see its own windows' contexts."""
    ctx = ba.Context(0)
    path = ol.GOLDEN + "/PTH2.bhmm"
    model = ol.Model(path, 0)
    hmm = ba.HMM(path, 0)
    rng = np.random.default_rng(7)
    genome = planted_genome(rng, model, L=300000)
    wins = bd.split_targets([len(genome)], hmm.max_length, 20000)
    seqs = [genome[s:s + n] for _, s, n, _ in wins]
    ctxs = [c for _, _, _, c in wins]
    om = ba.OProfile(ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
    out = []
    for k in (1, lanes):
        monkeypatch.setenv("BATH_HIP_LANES", str(k))
        block = ba.SeqBlock(ctx, seqs)
        block.set_context(ctxs)
        stats, res = pipe.run(block)
        out.append(((stats.nres, stats.n_past_msv, stats.pos_past_msv, stats.pos_past_bias, stats.pos_past_vit, stats.pos_past_fwd), np.sort(res, order=["window", "strand", "frame", "start"])))
    assert out[0][0] == out[1][0] and len(out[0][1]) == len(out[1][1])
    for f in out[0][1].dtype.names:                       # field by field: the records carry padding bytes
        assert np.array_equal(out[0][1][f], out[1][1][f], equal_nan=True), f
    pli, _, _, _ = model.run_pipeline_hits(seqs, contexts=ctxs)
    assert out[0][0] == (pli.nres, pli.n_past_msv, pli.pos_past_msv, pli.pos_past_bias, pli.pos_past_vit, pli.pos_past_fwd)
    assert pli.nres == 2 * len(genome)


def _records(dm, lo=0):
    return sorted((int(d.window) + lo, int(d.strand), d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm, int(d.reported),
                   int(np.float32(d.envsc).view(np.uint32)), int(np.float32(d.bitscore).view(np.uint32)), float(d.lnP)) for d in dm)


@pytest.mark.parametrize("hmmfile,fs", [("PTH2.bhmm", False), ("Caudal_act.bhmm", False), ("Caudal_act.bhmm", True)])
def test_blocks_of_a_search_report_what_the_whole_search_reports(hmmfile, fs):
    """The early E-value test of the domain stage uses pli->nres as it stands when the hit's window and strand are searched
    (p7_pipeline.c:1246, p7_domaindef.c:1033; bathsearch.c:1071 / :1084 count a window's W before each strand): a hit that is
    reported in the search's first windows can be dropped in its last.  bath_pipeline_params.nres_before tells a block where in the
    search it starts, so a search cut into blocks (ranks, worker contexts: bench.py's configs[3] / configs[4] legs) gives record for
    record -- the `reported` flag included -- the domains of the search run as one block, and both agree with the oracle's serial loop.
    The inputs hold hits on both sides of the threshold: many weak planted fragments, E_report small enough to cut through them."""
    ctx = ba.Context(0)
    path = ol.GOLDEN + "/" + hmmfile
    model = ol.Model(path, 0)
    hmm = ba.HMM(path, 0)
    rng = np.random.default_rng(99)
    L = 300000
    g = rng.integers(0, 4, size=L).astype(np.uint8)
    frags = common.emit_from_model(rng, model, 60, flank=1, sharpen=1.5)
    for aa, p in zip(frags, rng.integers(500, L - 3000, size=len(frags))):
        aa = aa[: max(12, len(aa) // int(rng.integers(1, 5)))]                      # whole domains and weak pieces of them
        nt = np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8)
        if fs and len(nt) > 60 and rng.integers(0, 2):                              # a frameshift: one nucleotide lost mid-gene
            nt = np.delete(nt, len(nt) // 2 + int(rng.integers(-9, 10)))
        if rng.integers(0, 2):
            nt = (3 - nt[::-1]).astype(np.uint8)
        g[p:p + len(nt)] = nt
    wins = bd.split_targets([L], hmm.max_length, 20000)
    seqs = [g[s:s + n] for _, s, n, _ in wins]
    ctxs = [c for _, _, _, c in wins]
    om = ba.OProfile(ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(ctx, om, fs_pipe=fs, ncbi_table=hmm.ct)
    om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct)) if fs else None
    om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct)) if fs else None
    E = 1e-3 if hmmfile == "PTH2.bhmm" else 1e-4

    def run(lo, hi, before):
        blk = ba.SeqBlock(ctx, seqs[lo:hi]); blk.set_context(ctxs[lo:hi])
        if fs:
            st, _, dm, _ = pipe.run_frameshift_domains(om3, om5, blk, E_report=E, nres_before=before)
        else:
            st, dm, _ = pipe.run_hits(blk, E_report=E, nres_before=before)
        return st, _records(dm, lo)

    st, whole = run(0, len(wins), 0)
    n_rep = sum(r[8] for r in whole)
    assert len(whole) >= 20 and 5 <= n_rep <= len(whole) - 5, (len(whole), n_rep)                 # hits on both sides of the threshold
    for G in (2, 5):
        parts, nres = [], 0
        for k in range(G):
            lo, hi = bd.shard_range(len(wins), k, G)
            stp, recs = run(lo, hi, 2 * sum(n - c for _, _, n, c in wins[:lo]))
            parts += recs; nres += int(stp.nres)
        assert nres == int(st.nres) and sorted(parts) == whole, G
    # the hit table of the search does not depend on the cut either, nor on the order in which the blocks' hits arrive (rank 0 of an
    # N-rank search gathers them in rank order: bench.py's configs[3] leg)
    if not fs:
        def table(arrs):
            h = ba.HitArray.concat(arrs)
            off = np.array([w[1] for w in wins], dtype=np.int64)[h.rec["window"]].astype(h.rec["ienv"].dtype)
            for f in ("ienv", "jenv", "iali", "jali"):
                h.rec[f] += off
            h.rec["window"] = 0
            th = ba.TopHits()
            th.add_arrays(h, ["chr"], [L])
            th.finalize(int(st.nres), hmm.max_length)
            return th.reported(), th.tblout(hmm.name, hmm.acc, hmm.M, show_cigar=True, show_header=False)

        def arrays(lo, hi):
            blk = ba.SeqBlock(ctx, seqs[lo:hi]); blk.set_context(ctxs[lo:hi])
            _, a, _ = pipe.run_hits(blk, E_report=E, arrays=True, nres_before=2 * sum(n - c for _, _, n, c in wins[:lo]))
            a.rec["window"] += lo
            return a

        t_whole = table([arrays(0, len(wins))])
        cuts = [bd.shard_range(len(wins), k, 3) for k in range(3)]
        assert table([arrays(lo, hi) for lo, hi in cuts]) == t_whole and table([arrays(lo, hi) for lo, hi in cuts[::-1]]) == t_whole
        assert t_whole[0] >= 5
    # without the offset the later blocks count from zero and keep more: the flag is what differs
    lo, hi = bd.shard_range(len(wins), 1, 2)
    _, late = run(lo, hi, 0)
    assert sum(r[8] for r in late) >= sum(r[8] for r in whole if r[0] >= lo)
    # the oracle's serial loop (one pipeline object, its running count): the same flags window by window
    if fs:
        _, _, _, odm, per_d, _ = model.run_pipeline_fsdom(seqs, contexts=ctxs, E=E)
    else:
        _, odm, per_d, _ = model.run_pipeline_hits(seqs, contexts=ctxs, E=E)
    want = sorted((w, o.ienv, o.jenv, o.iali, o.jali, o.ihmm, o.jhmm, int(o.reported)) for w, (a, b) in enumerate(per_d) for o in odm[a:b])
    got = sorted((r[0],) + r[2:9] for r in whole)
    assert got == want
