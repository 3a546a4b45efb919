"""BASELINE configs[3] in miniature: tutorial/tRNA-proteins.bhmm (12 query models, M = 56..459, the loop per query of
bathsearch.c:737) against ONE genome with planted genes of every model, cut into the reference's windows with context
(esl_sqio_ReadWindow, --block_length 262144, C = 3 * max_length of the current query) and dealt to two "ranks" the way
bath_amd.dist shards them -- GPU against the oracle, model by model, window by window, domain by domain; then the
per-rank hit lists are merged (dist.gather_domains' record format) into one table per query, which must equal the table
of the unsharded search."""
import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol
from bath_amd import dist as bd
from test_hits_gpu import compare_hits

pytestmark = pytest.mark.gpu

DB = ol.GOLDEN + "/tRNA-proteins.bhmm"
GENOME_NT = 640000            # three windows of <= 262144 new nucleotides


@pytest.fixture(scope="module")
def genome():
    """Background + two or three genes of each of the 12 families, both strands, some straddling the window boundaries."""
    rng = np.random.default_rng(12)
    g = rng.integers(0, 4, size=GENOME_NT).astype(np.uint8)
    n = ba.HMM.count(DB)
    assert n == 12
    spots = [262144 - 200, 2 * 262144 - 90, 262144 - 1500, 2 * 262144 + 40] + [int(x) for x in rng.integers(2000, GENOME_NT - 4000, size=3 * n)]
    planted = []
    k = 0
    for q in range(n):
        model = ol.Model(DB, q)
        for aa in common.emit_from_model(rng, model, 3, flank=3, sharpen=2.5):
            nt = np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8)
            if k % 2:
                nt = (3 - nt[::-1]).astype(np.uint8)
            p = spots[k % len(spots)]
            k += 1
            g[p:p + len(nt)] = nt[: GENOME_NT - p]
            planted.append((q, p, len(nt)))
    return g, planted


def test_database_has_the_twelve_models():
    names = [ba.HMM(DB, q).name for q in range(ba.HMM.count(DB))]
    assert names == ["ATE_N", "GlutR_N", "PTH2", "RtcB", "TGT", "Thg1", "Trm56", "tRNA-synt_1_2", "tRNA-synt_1c_C", "tRNA-synt_2d", "tRNA-Thr_ED", "TruB_C"]
    assert [ba.HMM(DB, q).M for q in range(12)] == [78, 152, 116, 459, 238, 131, 121, 185, 192, 247, 136, 56]


@pytest.mark.parametrize("q", range(12))
def test_every_query_model_over_sharded_windows(genome, q):
    g, planted = genome
    ctx = ba.Context(0)
    hmm = ba.HMM(DB, q)
    model = ol.Model(DB, q)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
    wins = bd.split_targets([len(g)], hmm.max_length)                  # default --block_length
    assert len(wins) == 3 and [c for _, _, _, c in wins] == [0, 3 * hmm.max_length, 3 * hmm.max_length]
    seqs = [g[s:s + n] for _, s, n, _ in wins]
    ctxs = [c for _, _, _, c in wins]

    # the oracle, window by window with the same contexts
    pli, odm, per_d, oskip = model.run_pipeline_hits(seqs, contexts=ctxs)

    # two ranks, each with its contiguous shard of the windows (dist.shard_range); no data-path exchange
    merged, nres, nskip = [], 0, 0
    counters = np.zeros(10, np.int64)
    for rank in range(2):
        lo, hi = bd.shard_range(len(wins), rank, 2)
        block = ba.SeqBlock(ctx, seqs[lo:hi])
        block.set_context(ctxs[lo:hi])
        st, dm, sk = pipe.run_hits(block)
        for d in dm:
            d.window += lo                                             # what gather_domains does with window_offset
        merged += dm
        nskip += sk
        counters += np.array([st.nres, st.n_orfs, st.n_past_msv, st.n_past_bias, st.n_past_vit, st.n_past_fwd,
                              st.pos_past_msv, st.pos_past_bias, st.pos_past_vit, st.pos_past_fwd])
    want = np.array([pli.nres, pli.n_orfs, pli.n_past_msv, pli.n_past_bias, pli.n_past_vit, pli.n_past_fwd,
                     pli.pos_past_msv, pli.pos_past_bias, pli.pos_past_vit, pli.pos_past_fwd])
    assert np.array_equal(counters, want)                              # p7_pipeline_Merge of the two ranks == one worker
    assert counters[0] == 2 * len(g)
    n = compare_hits(merged, odm, per_d, nskip, oskip)

    # the planted genes of THIS family that the oracle reports are reported here too (short fragments of a family may score
    # below the reporting threshold on both sides; other families' genes may or may not cross-hit)
    mine = [(p, ln) for qq, p, ln in planted if qq == q]

    def found_in(domains, window_of):
        hit = set()
        for p, ln in mine:
            for w, d in domains:
                off = wins[w][1]
                lo_, hi_ = min(d.iali, d.jali) + off, max(d.iali, d.jali) + off
                if d.reported and lo_ < p + ln and hi_ > p:
                    hit.add(p)
                    break
        return hit
    got_found = found_in([(d.window, d) for d in merged], None)
    want_found = found_in([(w, o) for w, (a, b) in enumerate(per_d) for o in odm[a:b]], None)
    assert got_found == want_found and len(got_found) >= 1 and n >= len(got_found)

    # ... and the sharded search's table is the unsharded one's (duplicates from the overlaps removed on "rank 0")
    def table(domains, nres):
        th = ba.TopHits()
        th.add(domains, ["genome"], [len(g)])
        th.finalize(int(nres), hmm.max_length)
        return th.tblout(hmm.name, hmm.acc, hmm.M, show_cigar=True)
    for d in merged:
        off = wins[d.window][1]
        d.ienv += off; d.jenv += off; d.iali += off; d.jali += off
        d.window = 0
    block = ba.SeqBlock(ctx, seqs)
    block.set_context(ctxs)
    st1, dm1, _ = pipe.run_hits(block)
    for d in dm1:
        off = wins[d.window][1]
        d.ienv += off; d.jenv += off; d.iali += off; d.jali += off
        d.window = 0
    assert table(merged, counters[0]) == table(dm1, st1.nres)


@pytest.mark.parametrize("q", [0, 5, 11])
def test_frameshift_pipeline_for_database_models(genome, q):
    """--fs for three of the twelve (M = 78, 131, 56) on the first 120 kb: windows, branches and hits against the oracle."""
    from test_fs_pipeline_gpu import compare_domains
    g, planted = genome
    ctx = ba.Context(0)
    hmm = ba.HMM(DB, q)
    model = ol.Model(DB, q)
    rng = np.random.default_rng(100 + q)
    seq = g[:120000].copy()
    for j, aa in enumerate(common.emit_from_model(rng, model, 4, flank=3, sharpen=2.5)):
        nt = list(common.revtranslate(rng, aa, model.basic))
        for _ in range(2):
            p = int(rng.integers(10, max(11, len(nt) - 10)))
            if rng.random() < 0.5:
                del nt[p]
            else:
                nt.insert(p, int(rng.integers(0, 4)))
        nt = np.array(nt, dtype=np.uint8)
        if j % 2:
            nt = (3 - nt[::-1]).astype(np.uint8)
        p = 5000 + 25000 * j
        seq[p:p + len(nt)] = nt
    om = ba.OProfile(ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct))
    om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    stats, fw, dm, nskip = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(ctx, [seq]))
    _, ofw, per_w, odm, per_d, oskip = model.run_pipeline_fsdom([seq])
    assert sorted((w.window, w.strand, w.n, w.length) for w in fw) == sorted((i, o.strand, o.n, o.length) for i, (a, b) in enumerate(per_w) for o in ofw[a:b])
    assert nskip == oskip
    assert compare_domains(model, dm, odm, per_d, nskip) >= 3
