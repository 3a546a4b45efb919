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


def _cluster_both(idx, i, j, k, m, nsamples, fs):
    a = lambda v: np.ascontiguousarray(v, np.int32)
    idx, i, j, k, m = a(idx), a(i), a(j), a(k), a(m)
    p32 = lambda v: v.ctypes.data_as(C.POINTER(C.c_int32))
    env_p = np.zeros(2 * 64, np.int32); n_p = C.c_int32(0)
    assert ba.lib().bath_selftest_cluster_segments(len(idx), p32(idx), p32(i), p32(j), p32(k), p32(m), nsamples, fs, p32(env_p), 64, C.byref(n_p)) == 0
    L_ = ol.lib()
    L_.bo_selftest_cluster_segments.argtypes = [C.c_int] + [C.POINTER(C.c_int32)] * 5 + [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    env_o = (C.c_int * 128)()
    n_o = L_.bo_selftest_cluster_segments(len(idx), p32(idx), p32(i), p32(j), p32(k), p32(m), nsamples, fs, env_o, 64)
    return [tuple(env_p[2 * e:2 * e + 2]) for e in range(n_p.value)], [(env_o[2 * e], env_o[2 * e + 1]) for e in range(n_o)]


@pytest.mark.parametrize("fs", [0, 1])
def test_identical_short_segments_are_singletons_as_in_the_reference(fs):
    """The model overlap of p7_spensemble's link rule is min(m) - max(k) WITHOUT + 1 (p7_spensemble.c:207, :244): a segment of four
    model nodes or fewer is not even linked to an identical copy of itself, so N copies are N singleton clusters -- never one
    significant cluster.  (Round 4's duplicate merging assumed linked(a, a); the advisor caught it.)"""
    n = 120                                                     # 60 % of 200 samples carry the same short segment
    short = _cluster_both(np.arange(n), [30] * n, [41] * n, [10] * n, [13] * n, 200, fs)       # m - k + 1 = 4: 3/4 < 0.8
    assert short[0] == short[1] == []
    longer = _cluster_both(np.arange(n), [30] * n, [44] * n, [10] * n, [14] * n, 200, fs)      # 5 nodes: 4/5 >= 0.8, linked to itself
    assert longer[0] == longer[1] == [(30, 44)]
    # mixed: copies of a short segment beside a real cluster; the short ones must not join or form anything
    idx = list(range(100)) + list(range(100))
    i = [30] * 100 + [200] * 100; j = [41] * 100 + [500] * 100; k = [10] * 100 + [5] * 100; m = [13] * 100 + [105] * 100
    mixed = _cluster_both(idx, i, j, k, m, 200, fs)
    assert mixed[0] == mixed[1] == [(200, 500)]


@pytest.mark.parametrize("seed", range(6))
def test_clustering_of_random_segment_sets_equals_the_all_pairs_search(seed):
    """The product clusters the DISTINCT segments and hands the components back; the oracle runs the reference's all-pairs search."""
    rng = np.random.default_rng(seed)
    idx, i, j, k, m = [], [], [], [], []
    protos = [(int(rng.integers(1, 300)), int(rng.integers(2, 12))) for _ in range(int(rng.integers(2, 6)))]
    for t in range(200):
        for (start, klen) in protos:
            if rng.random() < 0.7:
                kk = int(rng.integers(1, 40)); mm = kk + klen + int(rng.integers(-1, 2)) * (rng.random() < 0.3)
                mm = max(mm, kk)
                ii = start + int(rng.integers(0, 3)) * (rng.random() < 0.3); jj = ii + 3 * (mm - kk + 1) + int(rng.integers(-2, 3))
                idx.append(t); i.append(ii); j.append(max(jj, ii)); k.append(kk); m.append(mm)
    for fs in (0, 1):
        got, want = _cluster_both(idx, i, j, k, m, 200, fs)
        assert got == want
