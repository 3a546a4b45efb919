"""GPU parity: the batched HIP filters (through the C ABI) against the oracle, per target.

Integer filters (SSV/MSV, Viterbi) must be BIT-EXACT in score and status (the reference's own tests
demand 0.001 against the exact-emulation identity, msvfilter.c:648, vitfilter.c:680; integers give 0).
Forward and the bias filter are fp32: tolerance 1e-4 relative / 2e-4 absolute nats, written below.
"""
import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol

pytestmark = pytest.mark.gpu

MODELS = [("Caudal_act.bhmm", 0), ("PTH2.bhmm", 0), ("2OG-FeII_Oxy_3.bhmm", 0), ("MET-ct4.bhmm", 0), ("MET-ct4.bhmm", 1),
          ("synthetic:1024", 0), ("synthetic:700", 0), ("synthetic:7", 0)]   # M=458: 2 lanes/target; M=1024: 4 lanes/target


@pytest.fixture(scope="module", params=MODELS, ids=[m[0] + "#" + str(m[1]) for m in MODELS])
def setup(request, gpu_ctx, tmp_path_factory):
    name, idx = request.param
    if name.startswith("synthetic:"):
        M = int(name.split(":")[1])
        path = common.write_synthetic_bhmm(str(tmp_path_factory.mktemp("hmm") / ("s%d.bhmm" % M)), M, seed=M)
    else:
        path = ol.GOLDEN + "/" + name
    model = ol.Model(path, idx)
    hmm = ba.HMM(path, idx)
    gm = ba.Profile(hmm)
    om = ba.OProfile(gpu_ctx, gm)
    rng = np.random.default_rng(42)
    seqs = common.random_aa(rng, 600, 20, 400)
    seqs += common.emit_from_model(rng, model, 200)
    seqs += common.emit_from_model(rng, model, 60, sharpen=3.0)          # strong hits: overflow / J-state branches
    seqs += [np.zeros(1, np.uint8), np.full(3, 19, np.uint8), seqs[0][:1]]  # L=1 edge cases (msvfilter.c:722)
    seqs += common.random_aa(rng, 20, 1500, 3000)
    sq = ba.SeqBlock(gpu_ctx, seqs)
    return gpu_ctx, model, om, seqs, sq


def _same_scores(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_ssv_bit_exact(setup):
    ctx, model, om, seqs, sq = setup
    sc, st = ba.SSVFilter(ctx, om, sq)
    osc, ost = common.oracle_scores(model, seqs, "bo_ssvfilter")
    assert np.array_equal(st, ost)
    ok = ost == 0
    assert _same_scores(sc[ok], osc[ok])
    assert set(np.unique(ost)) >= {0}, "test set must exercise the OK branch"


@pytest.mark.parametrize("M", [16, 33, 64, 100, 152, 153, 200, 260, 304, 305, 416, 417, 600, 830, 1024, 1300, 1664])
def test_ssv_every_register_tiling(gpu_ctx, tmp_path, M):
    """The lane-per-target SSV kernels are instantiated per register count (NR = 16 .. 208 in steps of 4 / 16) and lanes per
    target (G = 1, 2, 4, 8), with hand-pipelined LDS reads and a VGPR bound that depends on NR: a sweep over model lengths
    that lands on the shapes' boundaries (152 | 153: one | two lanes with narrow tiles; 304 | 305: back to one lane with a wide
    tile; 416 | 417: one | two lanes per target; 1024: four).  A target's lanes are 64/G apart in the wave (SsvGroups).
    Both entry points: the standalone filter (ssv_lane_kernel) and the cascade (ssv_orf_kernel, through the survivors'
    MSV scores and the counters)."""
    path = common.write_synthetic_bhmm(str(tmp_path / ("s%d.bhmm" % M)), M, seed=M)
    model = ol.Model(path, 0)
    hmm = ba.HMM(path, 0)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    rng = np.random.default_rng(M)
    seqs = common.random_aa(rng, 150, 20, 300) + common.emit_from_model(rng, model, 60) + common.emit_from_model(rng, model, 20, sharpen=3.0)
    sc, st = ba.SSVFilter(gpu_ctx, om, ba.SeqBlock(gpu_ctx, seqs))
    osc, ost = common.oracle_scores(model, seqs, "bo_ssvfilter")
    assert np.array_equal(st, ost)
    ok = ost == 0
    assert ok.sum() > 0 and _same_scores(sc[ok], osc[ok])
    wins = [np.array(common.revtranslate(rng, aa, model.basic), dtype=np.uint8) for aa in common.emit_from_model(rng, model, 12, flank=10)]
    wins += common.random_dna(rng, 24, 900)
    stats, res = ba.Pipeline(gpu_ctx, om, fs_pipe=False).run(ba.SeqBlock(gpu_ctx, wins))
    pli, ores, _ = model.run_pipeline(wins)
    for f in ("n_orfs", "n_past_msv", "pos_past_msv", "pos_past_bias", "pos_past_vit", "pos_past_fwd"):
        assert getattr(stats, f) == getattr(pli, f), f
    want = sorted(np.float32(r.usc).view(np.uint32) for r in ores if r.stage >= 1)
    got = sorted(np.float32(u).view(np.uint32) for u in res["usc"])
    assert want == got
    if M == 1300:
        # above ~1100 nodes the Forward / Backward tables no longer fit the LDS and are read from global memory; above 1024 the
        # envelope kernels take the lane-per-envelope path: the hits must still be the oracle's
        _, dm, _ = ba.Pipeline(gpu_ctx, om, fs_pipe=False).run_hits(ba.SeqBlock(gpu_ctx, wins[:12]))
        _, odm, per_d, _ = model.run_pipeline_hits(wins[:12])
        key = lambda w, d: (w, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm)
        assert len(odm) >= 6 and sorted(key(d.window, d) for d in dm) == sorted(key(w, o) for w, (a, b) in enumerate(per_d) for o in odm[a:b])


@pytest.mark.parametrize("M", [40, 46, 54, 62, 70, 78, 86, 94, 102, 110, 118, 126, 134, 142, 150])
def test_msv_lane_every_register_tiling(gpu_ctx, tmp_path, M):
    """msv_lane_kernel<NR> (bath_msv_lane.hip: the J-state recurrence with a lane per target, models up to 152 nodes) is instantiated
    per register count NR = 16 .. 76 in steps of 4: every NR the other tests' models do not land on, score and status bit-exact,
    with targets that reach the J state, targets that overflow, single residues and lengths that are no multiple of the 8-residue
    reads."""
    path = common.write_synthetic_bhmm(str(tmp_path / ("s%d.bhmm" % M)), M, seed=M)
    model = ol.Model(path, 0)
    om = ba.OProfile(gpu_ctx, ba.Profile(ba.HMM(path, 0)))
    rng = np.random.default_rng(1000 + M)
    seqs = common.random_aa(rng, 120, 1, 97) + common.emit_from_model(rng, model, 80) + common.emit_from_model(rng, model, 40, sharpen=3.0)
    seqs += [np.concatenate(common.emit_from_model(rng, model, 3, sharpen=3.0)) for _ in range(6)]       # several domains in a row: J state, overflow
    sc, st = ba.MSVFilter(gpu_ctx, om, ba.SeqBlock(gpu_ctx, seqs))
    osc, ost = common.oracle_scores(model, seqs, "bo_msvfilter")
    assert np.array_equal(st, ost)
    assert _same_scores(sc, osc)
    _, sst = common.oracle_scores(model, seqs, "bo_ssvfilter")
    assert (sst != 0).sum() >= 5                             # targets SSV could not decide: they went through the lane kernel


def test_msv_bit_exact(setup):
    ctx, model, om, seqs, sq = setup
    sc, st = ba.MSVFilter(ctx, om, sq)
    osc, ost = common.oracle_scores(model, seqs, "bo_msvfilter")
    assert np.array_equal(st, ost)
    assert _same_scores(sc, osc)
    # the set has to reach the J-state re-run and the overflow branch, or this test proves little
    _, sst = common.oracle_scores(model, seqs, "bo_ssvfilter")
    if model.M >= 50:
        assert (sst == 19).sum() > 0 and (ost == 16).sum() > 0


def test_viterbi_bit_exact(setup):
    ctx, model, om, seqs, sq = setup
    sc, st = ba.ViterbiFilter(ctx, om, sq)
    osc, ost = common.oracle_scores(model, seqs, "bo_vitfilter")
    assert np.array_equal(st, ost)
    assert _same_scores(sc, osc)


def test_forward_parser(setup):
    ctx, model, om, seqs, sq = setup
    sc, st = ba.ForwardParser(ctx, om, sq)
    osc, ost = common.oracle_scores(model, seqs, "bo_forward_parser")
    assert np.array_equal(st, ost)
    assert np.allclose(sc, osc, rtol=1e-4, atol=2e-4)


def test_backward_parser(setup):
    """p7_BackwardParser on the GPU against the oracle (oracle/filters.c, fwdback.c:468-740): score, and every
    special-state row {E,N,J,B,C} of both passes with the same per-row SCALE.  Backward is rescaled by Forward's row
    factors, which are exactly 1 except on the sparse rows where xE passed 1e4, so the SCALE columns must be identical
    unless a row's xE sits within rounding of that threshold.  Forward == Backward per target (the reference's own
    utest, fwdback.c:1015-1085, demands 0.001)."""
    ctx, model, om, seqs, sq = setup
    fsc, bsc, fst, bst, fx, bx = ba.FwdBackParser(ctx, om, sq)
    L_ = ol.lib()
    import ctypes as C
    out = C.c_float(0)
    checked = 0
    for i, s_ in enumerate(seqs):
        n = len(s_)
        d = ol.dsq_from(s_)
        L_.bo_oprofile_reconfig_length(model.om, n)
        ofx = np.zeros((n + 1) * 6, np.float32); obx = np.zeros((n + 1) * 6, np.float32)
        st_f = L_.bo_forward_parser(ol.u8(d), n, model.om, ofx.ctypes.data_as(C.POINTER(C.c_float)), C.byref(out)); of = out.value
        st_b = L_.bo_backward_parser(ol.u8(d), n, model.om, ofx.ctypes.data_as(C.POINTER(C.c_float)), obx.ctypes.data_as(C.POINTER(C.c_float)), C.byref(out)); ob = out.value
        assert (fst[i], bst[i]) == (st_f, st_b)
        if st_f != 0 or st_b != 0:
            continue
        assert abs(fsc[i] - of) <= 2e-4 + 1e-4 * abs(of)
        assert abs(bsc[i] - ob) <= 2e-4 + 1e-4 * abs(ob)
        assert abs(fsc[i] - bsc[i]) <= 1e-3 + 1e-4 * abs(of)
        ofx, obx = ofx.reshape(-1, 6), obx.reshape(-1, 6)
        if i % 7 == 0 or n > 1000:                        # row-level comparison on a sample (and on every long target)
            same_scale = np.array_equal(fx[i][:, 5], ofx[:, 5])
            if same_scale:
                assert np.allclose(fx[i][:, :5], ofx[:, :5], rtol=2e-4, atol=1e-30)
                assert np.array_equal(bx[i][:, 5], obx[:, 5])
                assert np.allclose(bx[i][:, :5], obx[:, :5], rtol=5e-4, atol=1e-30)
                checked += 1
    assert checked > 50


def test_bias_filter(setup):
    ctx, model, om, seqs, sq = setup
    nullsc, fsc = ba.BiasFilter(ctx, om, sq)
    onull, ofsc = common.oracle_bias(model, seqs)
    assert _same_scores(nullsc, onull)
    assert np.allclose(fsc, ofsc, rtol=1e-5, atol=1e-4)


def test_empty_block(gpu_ctx, setup):
    ctx, model, om, seqs, sq = setup
    empty = ba.SeqBlock(ctx, [])
    sc, st = ba.MSVFilter(ctx, om, empty)
    assert sc.shape == (0,) and st.shape == (0,)
