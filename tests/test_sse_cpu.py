"""oracle/sse -- the SSE2 striped restatement of the reference's impl_sse filters (the CPU baseline bench.py times) -- against
the scalar oracle: the integer filters must agree bit for bit in score AND status (saturation, overflow, the SSV's "cannot
decide" outcome), the fp32 Forward parser to 1e-5 relative (same products, row sums accumulated in striped order).  The
cascade run on the striped kernels must give the scalar cascade's counters and records."""
import ctypes as C

import numpy as np
import pytest

import common

ENORESULT, ERANGE = 19, 16
import oracle_lib as ol

MODELS = [("Caudal_act.bhmm", 0), ("PTH2.bhmm", 0), ("AMP_N.bhmm", 0), ("MET-ct4.bhmm", 1), ("tRNA-proteins.bhmm", 3), ("tRNA-proteins.bhmm", 11)]


@pytest.fixture(scope="module", params=MODELS, ids=["%s-%d" % m for m in MODELS])
def setup(request):
    name, idx = request.param
    model = ol.Model(ol.GOLDEN + "/" + name, idx)
    rng = np.random.default_rng(5 + idx)
    seqs = common.random_aa(rng, 60, 20, 400) + common.emit_from_model(rng, model, 40) + common.emit_from_model(rng, model, 20, sharpen=3.0)
    seqs += [np.array([0], np.uint8), rng.choice(20, size=1, p=common.BG / common.BG.sum()).astype(np.uint8), common.random_aa(rng, 1, 2000, 2000)[0]]
    so = ol.lib().bs_oprofile_create(model.om)
    yield model, seqs, so
    ol.lib().bs_oprofile_free(so)


def both(model, so, seqs, scalar, striped):
    L_ = ol.lib()
    a, b = C.c_float(), C.c_float()
    out = []
    for s in seqs:
        d = ol.dsq_from(s)
        L_.bo_oprofile_reconfig_length(model.om, len(s))
        a.value = b.value = 0.0
        sa = scalar(ol.u8(d), len(s), model.om, C.byref(a)) if scalar is not L_.bo_forward_parser else scalar(ol.u8(d), len(s), model.om, None, C.byref(a))
        sb = striped(ol.u8(d), len(s), so, C.byref(b))
        out.append((sa, a.value, sb, b.value))
    return out


@pytest.mark.parametrize("which", ["ssv", "msv", "vit"])
def test_integer_filters_bit_exact(setup, which):
    model, seqs, so = setup
    L_ = ol.lib()
    scalar, striped = {"ssv": (L_.bo_ssvfilter, L_.bs_ssvfilter), "msv": (L_.bo_msvfilter, L_.bs_msvfilter), "vit": (L_.bo_vitfilter, L_.bs_vitfilter)}[which]
    res = both(model, so, seqs, scalar, striped)
    statuses = set()
    for sa, a, sb, b in res:
        assert sa == sb
        statuses.add(sa)
        if sa != ENORESULT:
            assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), (a, b)
    assert 0 in statuses
    if which == "ssv" and model.M > 60:
        assert statuses & {ENORESULT, ERANGE}           # the homologs reach the J-state / overflow outcomes


def test_msv_with_j_state_bit_exact(setup):
    """The full byte recurrence (msvfilter.c:106-207), not only the SSV shortcut: repeat-carrying sequences use the J state."""
    model, seqs, so = setup
    L_ = ol.lib()
    L_.bs_msv_full.argtypes = L_.bs_msvfilter.argtypes
    for sa, a, sb, b in both(model, so, seqs, L_.bo_msvfilter_noSSV, L_.bs_msv_full):
        assert sa == sb and np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32)


def test_forward_parser(setup):
    model, seqs, so = setup
    L_ = ol.lib()
    for sa, a, sb, b in both(model, so, seqs, L_.bo_forward_parser, L_.bs_forward_parser):
        assert sa == sb
        assert (np.isinf(a) and np.isinf(b)) or abs(a - b) <= 1e-5 * max(1.0, abs(a)), (a, b)


def test_viterbi_windows(setup):
    """p7_ViterbiFilter_BATH's hit windows (vitfilter.c:386-424) from the striped kernel."""
    model, seqs, so = setup
    L_ = ol.lib()
    n_win = 0
    for s in seqs:
        d = ol.dsq_from(s)
        L_.bo_oprofile_reconfig_length(model.om, len(s))
        outs = []
        for fn, prof in ((L_.bo_vitfilter_bath, model.om), (L_.bs_vitfilter_bath, so)):
            wl = ol.WindowList(); L_.bo_windowlist_init(C.byref(wl))
            sc = C.c_float()
            st = fn(ol.u8(d), len(s), prof, model.sd, C.c_float(-5.0), C.c_double(1e-3), C.byref(wl), C.byref(sc))
            outs.append((st, np.float32(sc.value).view(np.uint32), [(wl.w[i].n, wl.w[i].k, wl.w[i].length) for i in range(wl.count)]))
            L_.bo_windowlist_free(C.byref(wl))
        assert outs[0] == outs[1]
        n_win += len(outs[0][2])
    assert n_win > 0


def test_cascade_on_striped_kernels_equals_scalar_cascade():
    model = ol.Model(ol.GOLDEN + "/Caudal_act.bhmm", 0)
    rng = np.random.default_rng(3)
    wins = common.random_dna(rng, 150, 1000)
    for aa in common.emit_from_model(rng, model, 40, flank=10):
        wins.append(common.revtranslate(rng, aa, model.basic))
    L_ = ol.lib()
    out = []
    for on in (0, 1):
        L_.bo_pipeline_use_sse(on)
        pli, res, _ = model.run_pipeline(wins)
        out.append(([getattr(pli, f) for f in ("nres", "n_orfs", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd", "pos_past_msv", "pos_past_bias",
                                               "pos_past_vit", "pos_past_fwd", "cells_msv", "cells_vit", "cells_fwd")],
                    [(r.strand, r.frame, r.start, r.n, r.stage, r.msv_status, np.float32(r.usc).view(np.uint32), np.float32(r.vfsc).view(np.uint32)) for r in res],
                    [r.fwdsc for r in res]))
    L_.bo_pipeline_use_sse(0)
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1] and out[0][0][5] >= 10
    for a, b in zip(out[0][2], out[1][2]):
        assert (np.isinf(a) and np.isinf(b)) or abs(a - b) <= 1e-5 * max(1.0, abs(a))


@pytest.mark.parametrize("name,idx", [("Caudal_act.bhmm", 0), ("PTH2.bhmm", 0), ("AMP_N.bhmm", 0), ("tRNA-proteins.bhmm", 3)])
def test_fs3_forward_parser_striped_probability_space(name, idx):
    """oracle/sse/sse_fs.c -- the SSE2 striped, probability-space restatement of p7_ForwardParser_Frameshift_3Codons
    (impl_sse/fwdback_fs.c:97-533) that the --fs cpu_baseline legs run -- against the scalar log-space oracle
    (generic_fwdback_frameshift.c:451-622 restated) on DNA windows with planted frameshifted genes, random DNA, degenerate
    nucleotides and very short windows.  Tolerances are the reference's own between its SSE and generic parsers
    (fwdback_fs.c:3189-3191): with the oracle on EXACT log-sums the scores agree to 1e-4 relative (+1e-3 nats near zero) and so do
    all five special-state rows; against the TABLE log-sum (what the pipeline's oracle runs) within 1e-2 nats + 1e-4 relative per
    window -- the table's own distance from exact arithmetic (a truncating 0.001-nat table is biased; the reference allows 1.0)."""
    L_ = ol.lib()
    model = ol.Model(ol.GOLDEN + "/" + name, idx)
    gm3 = model.fs(3)
    L_.bs_fsprofile_create.restype = C.c_void_p
    L_.bs_fs3_forward_parser.restype = C.c_int
    L_.bs_fs3_forward_parser.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    L_.bs_fsprofile_free.argtypes = [C.c_void_p]
    L_.bs_fs3_backward_parser.restype = C.c_int
    L_.bs_fs3_backward_parser.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    so = L_.bs_fsprofile_create(gm3)
    assert so
    rng = np.random.default_rng(11 + idx)
    wins = []
    for aa in common.emit_from_model(rng, model, 10, flank=8):                     # genes with a deleted and an inserted nucleotide
        nt = list(common.revtranslate(rng, aa, model.basic))
        del nt[int(rng.integers(5, len(nt) - 5))]
        nt.insert(int(rng.integers(5, len(nt) - 5)), int(rng.integers(0, 4)))
        wins.append(np.array(nt, dtype=np.uint8))
    wins += [w for w in common.random_dna(rng, 6, 700)]
    deg = common.random_dna(rng, 1, 300)[0].copy(); deg[50:60] = 15; deg[120] = 4      # N runs and an ambiguity code
    wins += [deg, common.random_dna(rng, 1, 3)[0], common.random_dna(rng, 1, 7)[0], common.random_dna(rng, 1, 2500)[0]]
    f, g = C.c_float(), C.c_float()
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    worst_exact = worst_table = 0.0
    for w in wins:
        L = len(w)
        d = ol.u8(ol.dsq_from(w))
        L_.bo_fs_profile_reconfig_length(gm3, L // 3)
        rows = np.zeros((L + 1) * 5, np.float32)
        st = L_.bs_fs3_forward_parser(C.cast(d, C.c_void_p), L, so, rows.ctypes.data, C.byref(g))
        gx = L_.bo_gmx_create(model.M, L + 1, L, 3)
        for exact in (1, 0):
            L_.bo_flogsum_set_exact(exact)
            so_st = L_.bo_gforward_parser_fs3(d, L, gm3, gx, C.byref(f))
            assert so_st == 0
            if exact:
                orow = np.ctypeslib.as_array(gx.contents.xmx, shape=((L + 1) * 5,)).copy()
        L_.bo_flogsum_set_exact(0)
        table_sc = f.value
        L_.bo_flogsum_set_exact(1); L_.bo_gforward_parser_fs3(d, L, gm3, gx, C.byref(f)); L_.bo_flogsum_set_exact(0)
        exact_sc = f.value
        # ... and the Backward parser: score (= Forward's, up to the arithmetic) and rows against the exact-log-sum oracle
        L_.bo_flogsum_set_exact(1)
        b_st = L_.bo_gbackward_parser_fs3(d, L, gm3, gx, C.byref(f))
        L_.bo_flogsum_set_exact(0)
        b_exact = f.value
        brow_o = np.ctypeslib.as_array(gx.contents.xmx, shape=((L + 1) * 5,)).copy()
        L_.bo_gmx_free(gx)
        brow = np.zeros((L + 1) * 5, np.float32)
        h = C.c_float()
        bst = L_.bs_fs3_backward_parser(C.cast(d, C.c_void_p), L, so, brow.ctypes.data, C.byref(h))
        if b_st == 0 and np.isfinite(b_exact):
            assert bst == 0 and abs(h.value - b_exact) <= 1e-3 + 1e-4 * abs(b_exact), (L, h.value, b_exact)
            assert abs(h.value - g.value) <= 2e-3 + 2e-4 * abs(g.value) or not np.isfinite(exact_sc)        # Forward == Backward
            o2, s2 = brow_o.reshape(L + 1, 5), brow.reshape(L + 1, 5)
            top = np.where(np.isfinite(o2), o2, -np.inf).max(axis=1, keepdims=True)
            live = np.isfinite(o2) & (o2 > top - 40.0)                                   # within 40 nats of the row's largest value: what a posterior can see
            assert np.all(np.abs(s2[live] - o2[live]) <= 2e-3 + 2e-4 * np.abs(o2[live])), (L, float(np.abs(s2[live] - o2[live]).max()))
        if not np.isfinite(exact_sc):
            assert st == ERANGE or not np.isfinite(g.value)
            continue
        assert st == 0
        assert abs(g.value - exact_sc) <= 1e-3 + 1e-4 * abs(exact_sc), (L, g.value, exact_sc)
        assert abs(g.value - table_sc) <= 1e-2 + 1e-4 * abs(table_sc), (L, g.value, table_sc)
        worst_exact = max(worst_exact, abs(g.value - exact_sc)); worst_table = max(worst_table, abs(g.value - table_sc))
        fin = np.isfinite(orow) & (orow > -60.0)                                       # (values far below the scale are flushed in odds-ratio space)
        assert np.all(np.abs(rows[fin] - orow[fin]) <= 2e-3 + 2e-4 * np.abs(orow[fin])), (L, float(np.abs(rows[fin] - orow[fin]).max()))
        assert np.all(rows[~np.isfinite(orow)] < -50.0) or np.all(~np.isfinite(rows[~np.isfinite(orow)]))
    L_.bs_fsprofile_free(so)
    print("fs3 striped vs scalar: worst |delta| %.2e nats (exact log-sums), %.2e (table)" % (worst_exact, worst_table))


def test_fs_pipeline_on_the_striped_parser_finds_the_same_windows():
    """The oracle's --fs pipeline with the striped parser in place (bo_fs_use_sse: what the fs / c5 cpu_baseline legs time): same DNA
    windows, Forward scores within the tolerance above, the same branch for every window whose decision is not within that tolerance
    of flipping."""
    L_ = ol.lib()
    model = ol.Model(ol.GOLDEN + "/Caudal_act.bhmm", 0)
    rng = np.random.default_rng(3)
    wins = list(common.random_dna(rng, 40, 1000))
    for aa in common.emit_from_model(rng, model, 20, flank=10):
        nt = list(common.revtranslate(rng, aa, model.basic))
        del nt[int(rng.integers(10, len(nt) - 10))]
        wins.append(np.array(nt, dtype=np.uint8))
    pli0, _, _, fw0, pw0 = model.run_pipeline_fs(wins)
    L_.bo_fs_use_sse(1)
    try:
        pli1, _, _, fw1, pw1 = model.run_pipeline_fs(wins)
    finally:
        L_.bo_fs_use_sse(0)
    assert pw0 == pw1 and len(fw0) == len(fw1) >= 5
    flips = 0
    for a, b in zip(fw0, fw1):
        assert (a.strand, a.n, a.length, a.orf_cnt) == (b.strand, b.n, b.length, b.orf_cnt)
        assert abs(a.fwdsc - b.fwdsc) <= 1e-2 + 1e-4 * abs(a.fwdsc)
        flips += a.branch != b.branch
    assert flips <= 1
    # ... and through domain definition (both parsers striped, their rows read by the log-space domain decoding): the same domains
    _, _, _, dm0, pd0, sk0 = model.run_pipeline_fsdom(wins)
    L_.bo_fs_use_sse(1)
    try:
        _, _, _, dm1, pd1, sk1 = model.run_pipeline_fsdom(wins)
    finally:
        L_.bo_fs_use_sse(0)
    key = lambda d: (d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm)
    assert len(dm0) == len(dm1) >= 3 and sk0 == sk1
    assert sum(key(a) == key(b) for a, b in zip(dm0, dm1)) >= len(dm0) - 1          # (an envelope end may move where a posterior sits on a threshold)
    assert all(abs(a.bitscore - b.bitscore) <= 0.05 for a, b in zip(dm0, dm1) if key(a) == key(b))
