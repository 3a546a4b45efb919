"""The per-domain traces the batched pipeline exports (bath_hip_domain_traces: P7_DOMAIN.tr with posteriors, p7_domaindef.c:1171,
:1330) on the GPU:

  * rendered with bath_alidisplay_print they reproduce EVERY alignment block of the reference's recorded runs byte for byte --
    codon row with its frameshift marks, translation row, match row, frame row, a posterior-probability digit per column
    (tests/golden/PTH2.out, AMP_N.out, MET-ct4.out, AMP_N-fs.out, AMP_N-frameline.out; tests/test_alidisplay_cpu.py does the
    same with the oracle's traces);
  * on planted inputs they equal the oracle's traces state for state (st, k, i, c) in both branches and on both strands, with
    the posteriors inside the tolerance of the posterior matrices (2e-5 strict frameshift branch, 1e-3 standard branch)."""
import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol
import test_alidisplay_cpu as A
import test_fs_pipeline_gpu as P

pytestmark = pytest.mark.gpu


def gpu_run(ctx, path, idx, seqs, fs, **opts):
    ctx.set_fs_strict(True)
    hmm = ba.HMM(path, idx)
    om = ba.OProfile(ctx, ba.Profile(hmm))
    pipe = ba.Pipeline(ctx, om, fs_pipe=fs, ncbi_table=hmm.ct, **opts)
    blk = ba.SeqBlock(ctx, seqs)
    if fs:
        om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct)); om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
        stats, fw, dm, _ = pipe.run_frameshift_domains(om3, om5, blk)
    else:
        stats, dm, _ = pipe.run_hits(blk)
    traces = pipe.traces()
    assert len(traces) == len(dm)
    return hmm, dm, traces


@pytest.mark.parametrize("outfile,hmmfile,fasta,fs,frameline", A.RUNS)
def test_gpu_traces_reproduce_recorded_alignment_blocks(gpu_ctx, outfile, hmmfile, fasta, fs, frameline):
    want = A.recorded_blocks(outfile)
    recs = ol.read_fasta(ol.GOLDEN + "/" + fasta)
    seqs = [ol.digitize_dna(s) for _, s in recs]
    ncols = 0
    for q, blocks in enumerate(want):
        hmm, dm, traces = gpu_run(gpu_ctx, ol.GOLDEN + "/" + hmmfile, q, seqs, fs)
        gm, gm5 = ba.Profile(hmm), ba.FSProfile(hmm, 5, ncbi_table=hmm.ct)
        rep = [(d, t) for d, t in zip(dm, traces) if d.reported]
        assert len(rep) == len(blocks)
        by_ali = {(d.iali, d.jali): (d, t) for d, t in rep}
        for a, b, text in blocks:
            d, trace = by_ali[(a, b)]
            both = [seqs[d.window], A.strand_codes(seqs[d.window], True)]
            got = A.render(hmm, gm, gm5, trace, both, d, recs[d.window][0].split()[0], frameline)
            assert got == text, "\n" + got + "\n--- recorded ---\n" + text
            ncols += trace[0].N
    assert ncols >= 60


def compare_traces(dm, traces, odoms, pp_tol):
    """GPU domains / traces against the oracle's: every domain's trace state for state."""
    key = lambda w, d: (w, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm)
    omap = {key(w, o): o for w, o in odoms}
    n, kinds = 0, set()
    for d, (t, st, k, i, c, pp) in zip(dm, traces):
        o = omap.get(key(d.window, d))
        if o is None:
            continue
        ot, ost, ok_, oi, oc, opp = ol.trace_arrays(o.trace_idx)
        assert (t.N, t.win_start, t.orf_start, t.frameshift) == (ot.N, ot.win_start, ot.orf_start, ot.frameshift), (d.window, t.N, ot.N)
        assert np.array_equal(st, ost) and np.array_equal(k, ok_) and np.array_equal(i, oi) and np.array_equal(c, oc), (d.window,)
        assert np.abs(pp - opp).max() <= pp_tol, (d.window, float(np.abs(pp - opp).max()))
        assert st[0] == ba.T_M and st[-1] == ba.T_M and k[0] == d.ihmm and k[-1] == d.jhmm
        kinds |= set(int(x) for x in st)
        n += 1
    return n, kinds


def test_frameshift_branch_traces_equal_the_oracles(gpu_ctx):
    rng = np.random.default_rng(41)
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    model = ol.Model(path, 0)
    wins = P.frameshifted_windows(rng, model, n=24)
    hmm, dm, traces = gpu_run(gpu_ctx, path, 0, wins, True)
    ol.lib().bo_traces_reset()
    pli, ofw, per_w, odm, per_d, _ = model.run_pipeline_fsdom(wins)
    odoms = [(w, o) for w, (a, b) in enumerate(per_d) for o in odm[a:b]]
    fs_idx = [q for q, (t, *_r) in enumerate(traces) if t.frameshift]
    std_idx = [q for q, (t, *_r) in enumerate(traces) if not t.frameshift]
    n_fs, kinds = compare_traces([dm[q] for q in fs_idx], [traces[q] for q in fs_idx], odoms, 2e-5)
    n_std, _ = compare_traces([dm[q] for q in std_idx], [traces[q] for q in std_idx], odoms, 1e-3)
    assert n_fs == len(fs_idx) >= 3 and n_std >= 1 and kinds == {ba.T_M, ba.T_D, ba.T_I}
    # a quasi-codon somewhere: the planted indels are what the frameshift branch is for
    assert any((c[(st == ba.T_M)] != 3).any() for t, st, k, i, c, pp in [traces[q] for q in fs_idx])


@pytest.mark.parametrize("initiator", [ba.INIT_ANY, ba.INIT_AUG])
def test_standard_branch_traces_equal_the_oracles(gpu_ctx, initiator):
    path = ol.GOLDEN + "/PTH2.bhmm"
    model = ol.Model(path, 0)
    rng = np.random.default_rng(3)
    wins = []
    for n_, aa in enumerate(common.emit_from_model(rng, model, 16, flank=5, sharpen=1.5)):
        nt = np.array(common.revtranslate(rng, [10] + list(aa), model.basic), dtype=np.uint8)
        w = np.concatenate([rng.integers(0, 4, size=int(rng.integers(3, 200))).astype(np.uint8), nt, rng.integers(0, 4, size=int(rng.integers(0, 200))).astype(np.uint8)])
        wins.append((3 - w[::-1]).astype(np.uint8) if n_ % 2 else w)
    opts = {"initiator": initiator}
    hmm, dm, traces = gpu_run(gpu_ctx, path, 0, wins, False, **opts)
    ol.lib().bo_traces_reset()
    pli, odm, per_d, _ = model.run_pipeline_hits(wins, opts=opts)
    odoms = [(w, o) for w, (a, b) in enumerate(per_d) for o in odm[a:b]]
    n, kinds = compare_traces(dm, traces, odoms, 1e-3)
    assert n == len(dm) >= 6 and ba.T_M in kinds
    # the blocks render (the GPU's and the oracle's traces give the same text), M at an ORF's first codon under -m
    gm = ba.Profile(hmm)
    omap = {(w, o.iali, o.jali): o for w, o in odoms}
    for d, tr in zip(dm, traces):
        both = [wins[d.window], A.strand_codes(wins[d.window], True)]
        win = both[1 if d.iali > d.jali else 0][tr[0].win_start - 1:]
        a = ba.alidisplay_print(tr, win, hmm, d.iali, d.jali, "w%d" % d.window, gm=gm, ncbi_table=hmm.ct, initiator=initiator)
        o = omap[(d.window, d.iali, d.jali)]
        b = ba.alidisplay_print(ol.trace_arrays(o.trace_idx), win, hmm, d.iali, d.jali, "w%d" % d.window, gm=gm, ncbi_table=hmm.ct, initiator=initiator)
        assert a == b and " PP\n" in a


def test_trace_invariants_on_a_bench_sized_block(gpu_ctx):
    """Size-independent properties of the exported traces at the bench's shape (200 000 x 1 kb windows with 1 % planted frameshifted
    domains, ~1000 domains in both branches): every trace runs from a match state at (ihmm, first codon) to a match state at
    (jhmm, jali); nodes never decrease and advance by one per M / D column; M / I columns advance the nucleotide position by
    their codon length (3 for I) -- so the columns account for the alignment's whole nucleotide span --; D columns carry no
    posterior, M / I columns a probability; the shifted-codon count and the column count are the domain record's."""
    from bath_amd import synth
    ctx = gpu_ctx
    ctx.set_fs_strict(True)
    hmm = ba.HMM(ol.GOLDEN + "/Caudal_act.bhmm", 0)
    flat, offsets = synth.dna_windows(200_000, 1000, seed=4242, hmm=hmm, frameshift=True)[:2]
    om = ba.OProfile(ctx, ba.Profile(hmm))
    om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3, ncbi_table=hmm.ct)); om5 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 5, ncbi_table=hmm.ct))
    pipe = ba.Pipeline(ctx, om, fs_pipe=True, ncbi_table=hmm.ct)
    stats, fw, dm, _ = pipe.run_frameshift_domains(om3, om5, ba.SeqBlock(ctx, flat, offsets))
    traces = pipe.traces()
    assert len(traces) == len(dm) >= 500
    n_fs = n_std = n_quasi = 0
    for d, (t, st, k, i, c, pp) in zip(dm, traces):
        assert t.N == d.ali_columns == len(st) >= 1
        assert st[0] == ba.T_M and st[-1] == ba.T_M and k[0] == d.ihmm and k[-1] == d.jhmm
        isM, isD, isI = st == ba.T_M, st == ba.T_D, st == ba.T_I
        assert (isM | isD | isI).all()
        assert (np.diff(k) == (isM | isD)[1:].astype(np.int32)).all()                      # a node per M / D column, none for I
        step = np.where(isM, c, np.where(isI, 3, 0)).astype(np.int64)
        assert ((c >= 1) & (c <= 5))[isM].all() and (c[~isM] == 0).all()
        emit = isM | isI
        pos = i[emit].astype(np.int64)
        assert (np.diff(pos) == step[emit][1:]).all()                                      # each emitting column ends its codon's length further on
        span = abs(d.jali - d.iali) + 1
        assert step.sum() == span, (d.window, step.sum(), span)
        assert (pp[isD] == 0).all() and ((pp[emit] >= 0) & (pp[emit] <= 1.0 + 1e-5)).all()
        if t.frameshift:
            n_fs += 1
            assert int((c[isM] != 3).sum()) == d.n_shifted_codons
            n_quasi += d.n_shifted_codons
        else:
            n_std += 1
            assert (c[isM] == 3).all()
    assert n_fs >= 300 and n_std >= 100 and n_quasi >= 100
