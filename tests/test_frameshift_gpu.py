"""GPU parity of the frameshift kernels against the oracle's restatement of generic_*_frameshift.c.

Three modes, three bars.
  BATH_LOGSUM_TABLE_SERIAL ("strict"): the table-driven p7_FLogsum with the sums along the model in the reference's own
      serial order.  Every log-sum then has the reference's operands: scores AND every special-state row must be
      BIT-IDENTICAL to the oracle (relative error 0 <= the north star's 1e-4).  This is the parity gate.
  BATH_LOGSUM_TABLE (default, fast): the same table log-sums, but D(i,k) / E(i) / B(i) summed with wavefront scans, i.e. a
      different association of the same terms; with a 0.001-nat truncating table that moves a score by O(1e-3) nats.
      Bound: |gpu - oracle| <= 1e-4 * |oracle| + 5e-3 nats; the ACHIEVED maximum relative and absolute errors are measured,
      printed and written to gpurun_out/fs_parity_errors.json (a relative bound alone is meaningless for a table-quantised
      sum near zero; the reference's own SIMD-vs-generic tolerance is 1.0 nat, fwdback_fs.c:3189).
  BATH_LOGSUM_EXACT: the oracle is switched to exact log-sums too; bound 1e-4 relative + 1e-4.
"""
import ctypes as C

import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def fs_windows(rng, model, n, with_degenerate=True):
    """Reverse-translated model emissions with frameshift indels (+-1, +-2 nt) and in-frame stops, plus random DNA."""
    out = []
    for aa in common.emit_from_model(rng, model, n, flank=8):
        nt = list(common.revtranslate(rng, aa, model.basic))
        j = 6
        while j < len(nt) - 6:
            r = rng.random()
            if r < 0.010:
                del nt[j]
            elif r < 0.020:
                nt.insert(j, int(rng.integers(0, 4)))
            elif r < 0.025:
                del nt[j:j + 2]
            elif r < 0.030:
                nt[j:j] = [int(rng.integers(0, 4)), int(rng.integers(0, 4))]
            elif r < 0.032:
                nt[j:j + 3] = [3, 0, 0]          # TAA
            j += 3
        out.append(np.array(nt, dtype=np.uint8))
    out += common.random_dna(rng, max(2, n // 4), 240)
    if with_degenerate:
        out += common.random_dna(rng, 2, 150, degenerate_frac=0.03)
    out += [rng.integers(0, 4, size=L).astype(np.uint8) for L in (15, 16, 17, 47)]
    return out


@pytest.fixture(scope="module", params=["Caudal_act.bhmm", "2OG-FeII_Oxy_3.bhmm"])
def setup(request, gpu_ctx):
    path = ol.GOLDEN + "/" + request.param
    model = ol.Model(path)
    hmm = ba.HMM(path)
    om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3))
    om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5))
    rng = np.random.default_rng(123)
    wins = fs_windows(rng, model, 24)
    return gpu_ctx, model, om3, om5, wins, ba.SeqBlock(gpu_ctx, wins)


def oracle_fs3(model, wins, backward, exact=False):
    L_ = ol.lib()
    L_.bo_flogsum_set_exact(1 if exact else 0)
    gm3 = model.fs(3)
    L_.bo_fs_profile_reconfig_multihit(gm3, 100)
    sc, xm = [], []
    out = C.c_float()
    for w in wins:
        L = len(w)
        d = ol.dsq_from(w)
        L_.bo_fs_profile_reconfig_length(gm3, L // 3)
        gx = L_.bo_gmx_create(model.M, L + 1, L, 3)
        fn = L_.bo_gbackward_parser_fs3 if backward else L_.bo_gforward_parser_fs3
        st = fn(ol.u8(d), L, gm3, gx, C.byref(out))
        assert st == 0
        sc.append(out.value)
        xm.append(np.ctypeslib.as_array(gx.contents.xmx, shape=(L + 1, 5)).copy())
        L_.bo_gmx_free(gx)
    L_.bo_flogsum_set_exact(0)
    return np.array(sc, np.float32), xm


ERRORS = {}


def record_errors(name, got, want):
    """Achieved error of the fast mode against the oracle, kept for the report."""
    import json, os
    g, w = np.asarray(got, np.float64), np.asarray(want, np.float64)
    fin = np.isfinite(g) & np.isfinite(w)
    d = np.abs(g[fin] - w[fin])
    rel = d / np.maximum(np.abs(w[fin]), 1e-30)
    big = np.abs(w[fin]) >= 1.0                       # scores of at least one nat: where a relative error means something
    ERRORS[name] = {"n": int(fin.sum()), "max_abs_nats": float(d.max()) if d.size else 0.0, "max_rel": float(rel.max()) if rel.size else 0.0,
                    "max_rel_scores_above_1_nat": float(rel[big].max()) if big.any() else 0.0}
    print("fs parity", name, ERRORS[name])
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "fs_parity_errors.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        old.update(ERRORS)
        json.dump(old, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    return ERRORS[name]


def identical(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return a.shape == b.shape and bool(np.all(a.view(np.uint32) == b.view(np.uint32)))


def close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    return bool(np.all(both_inf | (np.abs(a - b) <= atol + rtol * np.abs(b))))


def test_fs3_parsers_strict_are_bit_identical(setup):
    """The parity gate of row a6: serial-order table log-sums reproduce generic_fwdback_frameshift.c:451,1422 bit for bit --
    both scores and every special-state row {E,N,J,B,C} that domain definition reads."""
    ctx, model, om3, om5, wins, blk = setup
    for backward, fn in ((False, ba.FS3ForwardParser), (True, ba.FS3BackwardParser)):
        sc, xm = fn(ctx, om3, blk, logsum=ba.LOGSUM_TABLE_SERIAL, want_xmx=True)
        osc, oxm = oracle_fs3(model, wins, backward=backward)
        assert identical(sc, osc), (backward, np.abs(sc - osc).max())
        for g, o in zip(xm, oxm):
            assert identical(g, o), backward


@pytest.mark.parametrize("mode,rtol,atol", [(ba.LOGSUM_TABLE, 1e-4, 5e-3), (ba.LOGSUM_EXACT, 1e-4, 1e-4)])
def test_fs3_forward_parser(setup, mode, rtol, atol, request):
    ctx, model, om3, om5, wins, blk = setup
    sc, xm = ba.FS3ForwardParser(ctx, om3, blk, logsum=mode, want_xmx=True)
    osc, oxm = oracle_fs3(model, wins, backward=False, exact=(mode == ba.LOGSUM_EXACT))
    if mode == ba.LOGSUM_TABLE:
        record_errors("fs3_forward/" + request.node.callspec.id, sc, osc)
    assert close(sc, osc, rtol, atol), np.abs(sc - osc).max()
    for g, o in zip(xm, oxm):                      # special-state rows feed domain definition (p7_domaindef.c:320)
        assert close(g[2:], o[2:], rtol, 4 * atol)


@pytest.mark.parametrize("mode,rtol,atol", [(ba.LOGSUM_TABLE, 1e-4, 5e-3), (ba.LOGSUM_EXACT, 1e-4, 1e-4)])
def test_fs3_backward_parser(setup, mode, rtol, atol, request):
    ctx, model, om3, om5, wins, blk = setup
    sc, xm = ba.FS3BackwardParser(ctx, om3, blk, logsum=mode, want_xmx=True)
    osc, oxm = oracle_fs3(model, wins, backward=True, exact=(mode == ba.LOGSUM_EXACT))
    if mode == ba.LOGSUM_TABLE:
        record_errors("fs3_backward/" + request.node.callspec.id, sc, osc)
    assert close(sc, osc, rtol, atol), np.abs(sc - osc).max()
    fsc = ba.FS3ForwardParser(ctx, om3, blk, logsum=mode)
    assert close(fsc, sc, 1e-4, 2e-2)              # Forward == Backward (generic_fwdback_frameshift.c:2304 unit test: 0.001 with exact sums)


def oracle_fs5(model, wins, c5_compat, exact=False):
    L_ = ol.lib()
    L_.bo_flogsum_set_exact(1 if exact else 0)
    gm5 = model.fs(5)
    res = []
    f, b, e = C.c_float(), C.c_float(), C.c_float()
    for w in wins:
        L = len(w)
        d = ol.dsq_from(w)
        L_.bo_fs_profile_reconfig_unihit(gm5, L // 3)           # p7_domaindef.c:324, :1020
        g8 = L_.bo_gmx_create(model.M, L + 1, L, 8)
        g3 = L_.bo_gmx_create(model.M, L + 1, L, 3)
        oa = L_.bo_gmx_create(model.M, L + 1, L, 3)
        assert L_.bo_gforward_fs(ol.u8(d), L, gm5, g8, 1 if c5_compat else 0, C.byref(f)) == 0
        assert L_.bo_gbackward_fs(ol.u8(d), L, gm5, g3, C.byref(b)) == 0
        L_.bo_gdecoding_fs(gm5, g8, g3)
        pp = np.ctypeslib.as_array(g8.contents.dp, shape=(L + 1, model.M + 1, 8)).copy()
        L_.bo_goptacc_fs(gm5, g8, oa, C.byref(e))
        oam = np.ctypeslib.as_array(oa.contents.dp, shape=(L + 1, model.M + 1, 3)).copy()
        n2 = np.zeros(29, np.float32)
        L_.bo_gnull2_fs(gm5, g8, ol.f32(n2))
        res.append((f.value, b.value, e.value, n2, pp, oam))
        for g in (g8, g3, oa):
            L_.bo_gmx_free(g)
    L_.bo_fs_profile_reconfig_multihit(gm5, 100)
    L_.bo_flogsum_set_exact(0)
    return res


def oa_matrices_agree(g, o, atol, rtol):
    """The optimal-accuracy matrix (generic_optacc_frameshift.c:53): max-sums of posteriors; -inf cells must coincide."""
    g, o = g[1:, 1:, :].astype(np.float64), o[1:, 1:, :].astype(np.float64)
    gi, oi = np.isneginf(g), np.isneginf(o)
    if not np.array_equal(gi, oi):
        return False
    g, o = np.where(gi, 0.0, g), np.where(oi, 0.0, o)
    return bool(np.all(np.abs(g - o) <= atol + rtol * np.abs(o)))


@pytest.mark.parametrize("c5_compat", [False, True])
@pytest.mark.parametrize("mode", [ba.LOGSUM_TABLE, ba.LOGSUM_TABLE_SERIAL], ids=["scan", "strict"])
def test_fs5_envelopes(setup, c5_compat, mode, request):
    ctx, model, om3, om5, wins, blk = setup
    env = [w for w in wins if len(w) >= 15]
    eb = ba.SeqBlock(ctx, env)
    got = ba.FS5Envelopes(ctx, om5, eb, logsum=mode, c5_compat=c5_compat, want_pp=True, want_oa=True)
    ref = oracle_fs5(model, env, c5_compat)
    fwd = np.array([r[0] for r in ref], np.float32); bwd = np.array([r[1] for r in ref], np.float32)
    strict = mode == ba.LOGSUM_TABLE_SERIAL
    if strict:                                      # row a7: p7_Forward_Frameshift / p7_Backward_Frameshift, bit for bit
        assert identical(got["fwdsc"], fwd) and identical(got["bcksc"], bwd)
    else:
        record_errors("fs5_forward/" + request.node.callspec.id, got["fwdsc"], fwd)
        record_errors("fs5_backward/" + request.node.callspec.id, got["bcksc"], bwd)
        assert close(got["fwdsc"], fwd, 1e-4, 5e-3), np.abs(got["fwdsc"] - fwd).max()
        assert close(got["bcksc"], bwd, 1e-4, 5e-3), np.abs(got["bcksc"] - bwd).max()
    # strict: Forward and Backward matrices are the oracle's, so posteriors differ only by expf's last bit and the division
    ptol, otol = (2e-5, 5e-4) if strict else (5e-3, 2e-2)
    for i, r in enumerate(ref):
        pp, oa = got["pp"][i], got["oa"][i]
        assert np.abs(pp[1:, 1:, 1:] - r[4][1:, 1:, 1:]).max() < ptol          # posteriors (decoding_fs.c:534 uses 0.001..0.2)
        assert abs(got["oasc"][i] - r[2]) < otol + 1e-3 * abs(r[2])            # expected # of correct positions
        assert oa_matrices_agree(oa, r[5], otol, 1e-3)                         # the whole OA matrix, not only its corner
        assert np.allclose(got["null2"][i], r[3], rtol=2e-3 if strict else 5e-3, atol=1e-4)        # null2_fs.c:193 uses 0.001..0.2; its log-sums run over column sums whose last bits differ (expf), so a table index may flip
    if not c5_compat:
        assert close(got["fwdsc"], got["bcksc"], 1e-4, 2e-2)                   # Forward == Backward


def test_fs_serial_switch_changes_nothing_but_the_order_of_the_launches(setup):
    """bath_hip_set_fs_serial (bench.py's fs.roofline.alone): Backward after Forward on one stream instead of beside it -- the same bits."""
    ctx, model, om3, om5, wins, blk = setup
    eb = ba.SeqBlock(ctx, [w for w in wins if len(w) >= 15])
    a = ba.FS5Envelopes(ctx, om5, eb, logsum=ba.LOGSUM_TABLE_SERIAL, want_oa=True)
    ctx.set_fs_serial(True)
    try:
        b = ba.FS5Envelopes(ctx, om5, eb, logsum=ba.LOGSUM_TABLE_SERIAL, want_oa=True)
    finally:
        ctx.set_fs_serial(None)
    for k in ("fwdsc", "bcksc", "oasc", "null2"):
        assert identical(a[k], b[k]), k
    assert all(identical(x, y) for x, y in zip(a["oa"], b["oa"]))


def test_fs5_multihit_forward_strict_is_bit_identical(setup):
    """p7_Forward_Frameshift in the MULTIHIT configuration of the model's saved length -- what p7_domaindef.c:411-414 runs on a
    multi-domain region before the stochastic tracebacks -- with strict log-sums: score, the whole matrix (8 cells per node) and
    the special-state rows must be the oracle's bit for bit, so that both sides draw the same samples."""
    ctx, model, om3, om5, wins, blk = setup
    env = [w for w in wins if len(w) >= 15]
    eb = ba.SeqBlock(ctx, env)
    M = model.M
    foff = np.zeros(len(env) + 1, np.int64); np.cumsum([(len(w) + 1) * (M + 1) * 8 for w in env], out=foff[1:])
    xoff = np.zeros(len(env) + 1, np.int64); np.cumsum([(len(w) + 1) * 5 for w in env], out=xoff[1:])
    sc = np.zeros(len(env), np.float32); fwd = np.zeros(int(foff[-1]), np.float32); xmx = np.zeros(int(xoff[-1]), np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ctx.set_fs_strict(True)
    try:
        ctx._check(ba.lib().bath_hip_fs5_forward_full(ctx._h, om5._h, eb._h, 100, fp(sc), fp(fwd), fp(xmx)), "fs5_forward_full")
    finally:
        ctx.set_fs_strict(False)
    L_ = ol.lib()
    gm5 = model.fs(5)
    L_.bo_fs_profile_reconfig_multihit(gm5, 100)
    f = C.c_float()
    for e, w in enumerate(env):
        L = len(w)
        g8 = L_.bo_gmx_create(M, L + 1, L, 8)
        assert L_.bo_gforward_fs(ol.u8(ol.dsq_from(w)), L, gm5, g8, 0, C.byref(f)) == 0
        dp = np.ctypeslib.as_array(g8.contents.dp, shape=(L + 1, M + 1, 8)).copy()
        ox = np.ctypeslib.as_array(g8.contents.xmx, shape=(L + 1, 5)).copy()
        L_.bo_gmx_free(g8)
        assert identical(np.float32(sc[e]), np.float32(f.value)), (e, sc[e], f.value)
        assert identical(xmx[xoff[e]:xoff[e + 1]].reshape(L + 1, 5), ox), e
        got = fwd[foff[e]:foff[e + 1]].reshape(L + 1, M + 1, 8)
        assert identical(got[1:, 1:, :], dp[1:, 1:, :]), e


def test_fs5_envelopes_with_the_multiwave_decode_oa_kernel():
    """fs5_decode_oa_mw_kernel (bath_fs_decode.hip: a block of waves per envelope, what models beyond 256 nodes get) forced onto the
    two small models with BATH_HIP_FS_OA_MW=1 (W = 2 waves at M = 145, one at M = 90): the posteriors, the whole optimal-accuracy
    matrix, its score and null2 against the oracle, same bars as the one-wave kernel.  Fresh process: the switch is read once."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, BATH_HIP_FS_OA_MW="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(here, "test_frameshift_gpu.py"), "-k", "test_fs5_envelopes and strict and not multiwave"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=os.path.dirname(here))
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
