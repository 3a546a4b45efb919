"""CPU tier: the frameshift branch's stochastic-trace ensemble as the PRODUCT runs it (bath_ensemble.hip is host code: 200
tracebacks from one random-number stream, memoised choice vectors, single-linkage clustering) against the oracle's restatement of
region_trace_ensemble_frameshift (p7_domaindef.c:891-958, generic_stotrace_frameshift.c:40-215) on the SAME multihit Forward
matrix -- the oracle's own p7_GForward_Frameshift of the region.  Identical matrices must give identical samples and therefore
identical envelopes: this is the host half of what tests/test_fs_strict_gpu.py::test_strict_pipeline_is_exact_on_clustered_regions
checks end to end on the GPU (there the matrix comes from fs5_fwd_chain_kernel, bit-identical to the oracle's)."""
import ctypes as C
import math
import time

import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol


def two_copy_windows(rng, model, n):
    """Two frameshifted copies of a gene a short spacer apart: the posterior profile of such a window is a multi-domain region."""
    genes = common.emit_from_model(rng, model, 2 * n, flank=3, sharpen=2.0)
    wins = []
    for a, b in zip(genes[::2], genes[1::2]):
        nt = [list(common.revtranslate(rng, g, model.basic)) for g in (a, b)]
        for seq in nt:
            p = int(rng.integers(10, len(seq) - 10))
            del seq[p]
        wins.append(np.array(nt[0] + list(rng.integers(0, 4, size=int(rng.integers(20, 60)))) + nt[1], dtype=np.uint8))
    return wins


@pytest.mark.parametrize("name", ["PTH2.bhmm", "Caudal_act.bhmm"])
def test_product_ensemble_equals_oracle_ensemble_on_the_oracles_matrix(name):
    path = ol.GOLDEN + "/" + name
    model = ol.Model(path, 0)
    hmm = ba.HMM(path, 0)
    tsc = ba.FSProfile(hmm, 5, ncbi_table=hmm.ct).arrays()[0].astype(np.float32)          # generic [M][8] log transitions
    M = model.M
    L_ = ol.lib()
    L_.bo_region_trace_ensemble_fs.restype = C.c_int
    gm5 = model.fs(5)
    L_.bo_fs_profile_reconfig_multihit(gm5, 100)                                            # the configuration bathsearch starts with (p7_domaindef.c:411-414)
    pm = (2.0 + 1.0) / (100.0 + 2.0 + 1.0)
    xNL, xNM, xE = math.log(1.0 - np.float32(pm)), math.log(np.float32(pm)), -0.69314718055994529
    rng = np.random.default_rng(7)
    f = C.c_float()
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    n_multi = 0
    t_prod = t_orac = 0.0
    for w in two_copy_windows(rng, model, 6):
        L = len(w)
        g8 = L_.bo_gmx_create(M, L + 1, L, 8)
        assert L_.bo_gforward_fs(ol.u8(ol.dsq_from(w)), L, gm5, g8, 0, C.byref(f)) == 0
        t0 = time.perf_counter()
        oenv = (C.c_int * 128)()
        on = L_.bo_region_trace_ensemble_fs(gm5, 1, L, g8, oenv, 64)
        t_orac += time.perf_counter() - t0
        fwd = np.ctypeslib.as_array(g8.contents.dp, shape=((L + 1) * (M + 1) * 8,)).astype(np.float32).copy()
        fx = np.ctypeslib.as_array(g8.contents.xmx, shape=((L + 1) * 5,)).astype(np.float32).copy()
        L_.bo_gmx_free(g8)
        env = np.zeros(128, np.int32)
        n = C.c_int32(0)
        t0 = time.perf_counter()
        st = ba.lib().bath_selftest_fs_ensemble(M, fp(tsc), xNL, xNM, xE, 1, L, fp(fwd), fp(fx), env.ctypes.data_as(C.POINTER(C.c_int32)), 64, C.byref(n))
        t_prod += time.perf_counter() - t0
        assert st == 0
        want = [(oenv[2 * e], oenv[2 * e + 1]) for e in range(on)]
        got = [(int(env[2 * e]), int(env[2 * e + 1])) for e in range(n.value)]
        assert got == want, (L, got, want)
        n_multi += len(got) >= 2
    assert n_multi >= 3                                     # the inputs do produce several clusters per region
    print("ensembles: product %.1f ms, oracle %.1f ms for 6 regions" % (t_prod * 1e3, t_orac * 1e3))
