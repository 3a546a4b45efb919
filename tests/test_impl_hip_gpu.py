"""impl_hip/ on the GPU: the single-target prototypes of impl_sse.h (p7_MSVFilter(dsq, L, om, ox, &sc), ...) called through the
test harness exactly as p7_Pipeline_BATH / p7_domaindef.c call them, against the batched C ABI on the same targets -- the
results must be the batched results bit for bit (it is the same kernels with n = 1) -- and against the oracle where the
shim adds host code of its own (domain decoding, the optimal-accuracy and stochastic tracebacks)."""
import ctypes as C

import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol

pytestmark = pytest.mark.gpu

f32p, i32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint8)


@pytest.fixture(scope="module")
def hs():
    import impl_hip_build
    L = C.CDLL(impl_hip_build.build())
    L.hs_filters.argtypes = [C.c_char_p, C.c_int, u8p, C.c_int, C.c_float, C.c_double, f32p, i32p, f32p, f32p, i32p, i32p, i32p, i32p, i32p, i32p, i32p, f32p, i32p]
    L.hs_std_envelope.argtypes = [C.c_char_p, C.c_int, u8p, C.c_int, f32p, f32p, C.c_char_p, i32p, i32p, f32p, C.c_int]
    L.hs_std_region.argtypes = [C.c_char_p, C.c_int, u8p, C.c_int, C.c_int, C.c_uint32, C.c_int, f32p, i32p, i32p, i32p]
    L.hs_fs_parsers.argtypes = [C.c_char_p, C.c_int, u8p, C.c_int, f32p, f32p, f32p, f32p, f32p, f32p]
    L.hs_std_decoding.argtypes = [C.c_char_p, C.c_int, u8p, C.c_int, f32p, f32p, f32p, f32p, f32p]
    L.hs_fs_envelope.argtypes = [C.c_char_p, C.c_int, u8p, C.c_int, f32p, f32p, C.c_char_p, i32p, i32p, i32p, f32p, C.c_int]
    L.hs_fs_region.argtypes = [C.c_char_p, C.c_int, u8p, C.c_int, C.c_uint32, C.c_int, f32p, i32p, i32p, i32p]
    return L


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def fp(a):
    return a.ctypes.data_as(f32p)


def ip(a):
    return a.ctypes.data_as(i32p)


PATH = ol.GOLDEN + "/Caudal_act.bhmm"
T_M, T_D, T_I, T_B, T_E = 1, 2, 3, 6, 7


def test_filters_and_parsers_equal_the_batched_calls(hs, gpu_ctx):
    model = ol.Model(PATH)
    hmm = ba.HMM(PATH)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    rng = np.random.default_rng(9)
    seqs = common.random_aa(rng, 6, 25, 300, with_degenerate=False) + common.emit_from_model(rng, model, 10) + common.emit_from_model(rng, model, 4, sharpen=3.0)
    blk = ba.SeqBlock(gpu_ctx, seqs)
    msv, msv_st = ba.MSVFilter(gpu_ctx, om, blk)
    vit, vit_st = ba.ViterbiFilter(gpu_ctx, om, blk)
    fsc, bsc, fst, bst, fx, bx = ba.FwdBackParser(gpu_ctx, om, blk)
    n_windows = 0
    for t, s in enumerate(seqs):
        L = len(s)
        d = ol.dsq_from(s)
        out, st = np.zeros(5, np.float32), np.zeros(5, np.int32)
        gx, gb = np.zeros((L + 1, 6), np.float32), np.zeros((L + 1, 6), np.float32)
        wn, wk, wl, nw = np.zeros(64, np.int32), np.zeros(64, np.int32), np.zeros(64, np.int32), np.zeros(1, np.int32)
        sn, sk, sl, ssc, snw = np.zeros(64, np.int32), np.zeros(64, np.int32), np.zeros(64, np.int32), np.zeros(64, np.float32), np.zeros(1, np.int32)
        filtersc = -5.0
        assert hs.hs_filters(PATH.encode(), 0, ol.u8(d), L, filtersc, 1e-3, fp(out), ip(st), fp(gx), fp(gb), ip(wn), ip(wk), ip(wl), ip(nw), ip(sn), ip(sk), ip(sl), fp(ssc), ip(snw)) == 0
        assert (st[0], st[1], st[3], st[4]) == (msv_st[t], vit_st[t], fst[t], bst[t])
        assert bits(out[0]) == bits(msv[t]) and bits(out[1]) == bits(vit[t]) and bits(out[2]) == bits(vit[t])
        assert bits(out[3]) == bits(fsc[t]) and bits(out[4]) == bits(bsc[t])
        assert np.array_equal(bits(gx), bits(fx[t])) and np.array_equal(bits(gb), bits(bx[t]))      # the rows p7_DomainDecoding reads
        # p7_ViterbiFilter_BATH / p7_SSVFilter_BATH windows against the oracle's (vitfilter.c:386-424, msvfilter.c:330-420)
        L_ = ol.lib()
        L_.bo_oprofile_reconfig_length(model.om, L)
        owl = ol.WindowList(); L_.bo_windowlist_init(C.byref(owl))
        osc = C.c_float()
        L_.bo_vitfilter_bath(ol.u8(d), L, model.om, model.sd, C.c_float(filtersc), C.c_double(1e-3), C.byref(owl), C.byref(osc))
        assert [(wn[i], wk[i], wl[i]) for i in range(nw[0])] == [(owl.w[i].n, owl.w[i].k, owl.w[i].length) for i in range(owl.count)]
        n_windows += owl.count
        L_.bo_windowlist_free(C.byref(owl))
        owl = ol.WindowList(); L_.bo_windowlist_init(C.byref(owl))
        L_.bo_ssvfilter_bath(ol.u8(d), L, model.om, model.sd, C.byref(model.bg), C.c_double(1e-3), C.byref(owl))
        assert [(sn[i], sk[i], sl[i]) for i in range(snw[0])] == [(owl.w[i].n, owl.w[i].k, owl.w[i].length) for i in range(owl.count)]
        assert all(abs(ssc[i] - owl.w[i].score) <= 1e-5 * max(1.0, abs(owl.w[i].score)) for i in range(owl.count))
        n_windows += owl.count
        L_.bo_windowlist_free(C.byref(owl))
    assert n_windows >= 10


def test_standard_envelope_call_sequence(hs, gpu_ctx):
    """p7_Forward -> p7_Backward -> p7_Decoding -> p7_OptimalAccuracy -> p7_OATrace -> p7_Null2_ByExpectation as
    rescore_isolated_domain_bath makes them: scores equal bath_hip_std_envelopes' (n = 1 of the same pass); the trace is a
    valid path whose match states carry the posteriors of the pass."""
    model = ol.Model(PATH)
    hmm = ba.HMM(PATH)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    rng = np.random.default_rng(4)
    seqs = common.emit_from_model(rng, model, 6, flank=4)
    blk = ba.SeqBlock(gpu_ctx, seqs)
    res = (ba.StdResult * len(seqs))()
    gpu_ctx._check(ba.lib().bath_hip_std_envelopes(gpu_ctx._h, om._h, blk._h, res, None, None, None, None), "std_envelopes")
    for t, s in enumerate(seqs):
        L = len(s)
        d = ol.dsq_from(s)
        out, null2 = np.zeros(3, np.float32), np.zeros(29, np.float32)
        cap = 4 * (L + hmm.M) + 64
        tst = C.create_string_buffer(cap)
        tk, ti, tpp = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        n = hs.hs_std_envelope(PATH.encode(), 0, ol.u8(d), L, fp(out), fp(null2), tst, ip(tk), ip(ti), fp(tpp), cap)
        assert n > 0
        assert bits(out[0]) == bits(res[t].fwdsc) and bits(out[1]) == bits(res[t].bcksc) and bits(out[2]) == bits(res[t].oasc)
        assert np.array_equal(bits(null2), bits(np.array(res[t].null2[:], np.float32)))
        assert abs(out[0] - out[1]) <= 1e-3 * max(1.0, abs(out[0]))                    # Forward == Backward
        st = [tst.raw[z] for z in range(n)]
        assert st[0] == 4 and st[-1] == 9 and st.count(T_B) == 1 and st.count(T_E) == 1     # S ... B core E ... T, one domain
        core = [z for z in range(n) if st[z] in (T_M, T_D, T_I)]
        ks = [tk[z] for z in core if st[z] != T_I]
        assert ks == list(range(ks[0], ks[-1] + 1))                                    # consecutive nodes
        emitted = [ti[z] for z in core if st[z] in (T_M, T_I)]
        assert emitted == list(range(emitted[0], emitted[-1] + 1))                     # consecutive residues
        # the expected number of correctly aligned residues (oasc) is what the traced states' posteriors add up to
        assert abs(sum(float(tpp[z]) for z in range(n)) - out[2]) <= 2e-3 * max(1.0, out[2])
        # ... and state by state it is the oracle's p7_OATrace (optacc.c:225-430, select_e in striped order) on its own matrices
        L_ = ol.lib()
        L_.bo_std_envelope_trace.argtypes = [C.POINTER(ol.OProfile), u8p, C.c_int, i32p, i32p, i32p, f32p]
        pst, pk, pi_ = np.zeros(L + hmm.M + 8, np.int32), np.zeros(L + hmm.M + 8, np.int32), np.zeros(L + hmm.M + 8, np.int32)
        ooa = np.zeros(1, np.float32)
        pn = L_.bo_std_envelope_trace(model.om, ol.u8(d), L, ip(pst), ip(pk), ip(pi_), fp(ooa))
        assert pn > 0 and abs(ooa[0] - out[2]) <= 1e-4 * max(1.0, out[2])
        o2p = {3: T_M, 4: T_D, 5: T_I}                                                  # BO_T_M/D/I -> p7T_M/D/I
        want = [(o2p[int(pst[z])], int(pk[z]), int(pi_[z])) for z in range(pn - 1, -1, -1)]
        first, last = core[0], max(z for z in core if st[z] == T_M)                     # first to last match state
        got_core = [(st[z], int(tk[z]), int(ti[z])) for z in range(first, last + 1)]
        assert [(a, b) for a, b, _ in got_core] == [(a, b) for a, b, _ in want]
        assert [c for a, _, c in got_core if a != T_D] == [c for a, _, c in want if a != T_D]


def test_standard_region_stochastic_traces(hs, gpu_ctx):
    """p7_oprofile_ReconfigMultihit(om, saveL); p7_Forward; 50 x p7_StochasticTrace from one generator (p7_domaindef.c:557-575):
    the Forward score is the batched multihit Forward's with that configuration length; traces are complete paths."""
    model = ol.Model(PATH)
    hmm = ba.HMM(PATH)
    om = ba.OProfile(gpu_ctx, ba.Profile(hmm))
    rng = np.random.default_rng(8)
    g = common.emit_from_model(rng, model, 2, flank=3, sharpen=2.0)
    s = np.concatenate([g[0], rng.choice(20, size=12).astype(np.uint8), g[1]])         # two domains in one region
    L, saveL = len(s), len(s) + 57
    d = ol.dsq_from(s)
    sc = np.zeros(1, np.float32)
    nd, fi, li = np.zeros(50, np.int32), np.zeros(50, np.int32), np.zeros(50, np.int32)
    assert hs.hs_std_region(PATH.encode(), 0, ol.u8(d), L, saveL, 42, 50, fp(sc), ip(nd), ip(fi), ip(li)) == 0
    blk = ba.SeqBlock(gpu_ctx, [s])
    bsc, bst = np.zeros(1, np.float32), np.zeros(1, np.int32)
    cfg = np.array([saveL], np.int32)
    gpu_ctx._check(ba.lib().bath_hip_forward_full(gpu_ctx._h, om._h, blk._h, ip(cfg), 0, fp(bsc), ip(bst), None, None), "forward_full")
    assert bits(sc[0]) == bits(bsc[0])
    assert nd.min() >= 2 and np.bincount(nd).max() >= 25                               # the samples agree on the region's domains (two genes, each possibly a repeat)
    assert fi.min() >= 1 and li.max() <= L


def test_standard_domain_decoding(hs, gpu_ctx):
    """p7_DomainDecoding (decoding.c:143-189) through the shim against the oracle's restatement fed with the shim's own parser
    rows: btot, etot and mocc are the same fp32 arithmetic in the same order."""
    model = ol.Model(PATH)
    rng = np.random.default_rng(17)
    seqs = common.emit_from_model(rng, model, 6, flank=6) + common.random_aa(rng, 3, 40, 200, with_degenerate=False)
    L_ = ol.lib()
    L_.bo_domain_decoding.argtypes = [C.POINTER(ol.OProfile), f32p, f32p, C.c_int, C.c_int, f32p, f32p, f32p]
    for s in seqs:
        L = len(s)
        d = ol.dsq_from(s)
        gx, gb = np.zeros((L + 1, 6), np.float32), np.zeros((L + 1, 6), np.float32)
        btot, etot, mocc = np.zeros(L + 1, np.float32), np.zeros(L + 1, np.float32), np.zeros(L + 1, np.float32)
        assert hs.hs_std_decoding(PATH.encode(), 0, ol.u8(d), L, fp(gx), fp(gb), fp(btot), fp(etot), fp(mocc)) == 0
        L_.bo_oprofile_reconfig_length(model.om, L)
        ob, oe, om_ = np.zeros(L + 1, np.float32), np.zeros(L + 1, np.float32), np.zeros(L + 1, np.float32)
        assert L_.bo_domain_decoding(model.om, fp(gx), fp(gb), L, 0, fp(ob), fp(oe), fp(om_)) == 0
        assert np.array_equal(bits(btot), bits(ob)) and np.array_equal(bits(etot), bits(oe)) and np.array_equal(bits(mocc), bits(om_))


def test_frameshift_parsers_and_domain_decoding(hs, gpu_ctx):
    model = ol.Model(PATH)
    hmm = ba.HMM(PATH)
    om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3))
    rng = np.random.default_rng(21)
    import test_frameshift_gpu as tf
    wins = tf.fs_windows(rng, model, 6, with_degenerate=False)[:8]
    blk = ba.SeqBlock(gpu_ctx, wins)
    fsc, fx = ba.FS3ForwardParser(gpu_ctx, om3, blk, logsum=ba.LOGSUM_CONTEXT, want_xmx=True)      # the mode the shims run in: the context's (strict)
    bsc, bx = ba.FS3BackwardParser(gpu_ctx, om3, blk, logsum=ba.LOGSUM_CONTEXT, want_xmx=True)
    L_ = ol.lib()
    for t, w in enumerate(wins):
        L = len(w)
        d = ol.dsq_from(w)
        out = np.zeros(2, np.float32)
        gx, gb = np.zeros((L + 1, 6), np.float32), np.zeros((L + 1, 6), np.float32)
        btot, etot, mocc = np.zeros(L + 1, np.float32), np.zeros(L + 1, np.float32), np.zeros(L + 1, np.float32)
        assert hs.hs_fs_parsers(PATH.encode(), 0, ol.u8(d), L, fp(out), fp(gx), fp(gb), fp(btot), fp(etot), fp(mocc)) == 0
        assert bits(out[0]) == bits(fsc[t]) and bits(out[1]) == bits(bsc[t])
        assert np.array_equal(bits(gx[:, :5]), bits(fx[t])) and np.array_equal(bits(gb[:, :5]), bits(bx[t]))
        # p7_DomainDecoding_Frameshift against the oracle's restatement of generic_decoding_frameshift.c:204 on the oracle's own rows
        # (identical to the shim's: the parsers run in the reference's serial order); what differs is expf's last bit
        gm3, gm5 = model.fs(3), model.fs(5)
        L_.bo_fs_profile_reconfig_multihit(gm3, 100); L_.bo_fs_profile_reconfig_length(gm3, L // 3)
        gf, gbk = L_.bo_gmx_create(model.M, L + 1, L, 3), L_.bo_gmx_create(model.M, L + 1, L, 3)
        o = C.c_float()
        assert L_.bo_gforward_parser_fs3(ol.u8(d), L, gm3, gf, C.byref(o)) == 0 and bits(np.float32(o.value)) == bits(out[0])
        assert L_.bo_gbackward_parser_fs3(ol.u8(d), L, gm3, gbk, C.byref(o)) == 0 and bits(np.float32(o.value)) == bits(out[1])
        L_.bo_fs_profile_reconfig_multihit(gm5, 100)                                     # the model's saved length (p7_domaindef.c:318)
        ob, oe, om_ = np.zeros(L + 1, np.float32), np.zeros(L + 1, np.float32), np.zeros(L + 1, np.float32)
        L_.bo_gdomain_decoding_fs.argtypes = [C.POINTER(ol.FsProfile), C.POINTER(ol.Gmx), C.POINTER(ol.Gmx), f32p, f32p, f32p]
        L_.bo_gdomain_decoding_fs(gm5, gf, gbk, fp(ob), fp(oe), fp(om_))
        L_.bo_gmx_free(gf); L_.bo_gmx_free(gbk)
        assert np.abs(btot - ob).max() <= 1e-5 * max(1.0, ob.max()) and np.abs(etot - oe).max() <= 1e-5 * max(1.0, oe.max())
        assert np.abs(mocc - om_).max() <= 2e-5
        assert np.all(np.diff(btot[::3]) >= -1e-6) and np.all(mocc[3:] <= 1.0 + 1e-5)
        if fsc[t] > 20.0:
            assert mocc.max() > 0.9 and btot.max() > 0.5 and etot.max() > 0.5           # a gene in the window: a domain begins and ends


def test_frameshift_envelope_call_sequence(hs, gpu_ctx):
    """rescore_isolated_domain_frameshift's calls (p7_domaindef.c:1019-1083) on the recorded AMP_N --fs hit: the envelope is the
    whole 411-nt target; the trace must be the alignment tutorial/AMP_N-fs.tbl records (hmm 1..131, ali 1..402, 6 shifted codons)."""
    path = ol.GOLDEN + "/AMP_N.bhmm"
    seq = ba.digitize(ol.read_fasta(ol.GOLDEN + "/target-AMP_N.fa")[0][1], ba.DNA_SYMS)
    hmm = ba.HMM(path)
    om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5))
    L = len(seq)
    d = ol.dsq_from(seq)
    out, null2 = np.zeros(3, np.float32), np.zeros(29, np.float32)
    cap = 4 * (L + hmm.M) + 64
    tst = C.create_string_buffer(cap)
    tk, ti, tc, tpp = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float32)
    n = hs.hs_fs_envelope(path.encode(), 0, ol.u8(d), L, fp(out), fp(null2), tst, ip(tk), ip(ti), ip(tc), fp(tpp), cap)
    assert n > 0
    got = ba.FS5Envelopes(gpu_ctx, om5, ba.SeqBlock(gpu_ctx, [seq]), logsum=ba.LOGSUM_CONTEXT)      # the shims run in the context's mode (strict)
    assert bits(out[0]) == bits(got["fwdsc"][0]) and bits(out[1]) == bits(got["bcksc"][0]) and bits(out[2]) == bits(got["oasc"][0])
    assert np.array_equal(bits(null2), bits(got["null2"][0]))
    st = [tst.raw[z] for z in range(n)]
    ms = [z for z in range(n) if st[z] == T_M]
    assert (tk[ms[0]], tk[ms[-1]]) == (1, 131)
    assert (ti[ms[0]] - (tc[ms[0]] - 1), ti[ms[-1]]) == (1, 402)
    assert sum(1 for z in ms if tc[z] != 3) == 6


class OTrace(C.Structure):
    """bo_trace (oracle/bath_oracle.h): P7_TRACE with codon lengths."""
    _fields_ = [("N", C.c_int), ("nalloc", C.c_int), ("st", C.POINTER(C.c_int8)), ("k", C.POINTER(C.c_int32)), ("i", C.POINTER(C.c_int32)),
                ("c", C.POINTER(C.c_int32)), ("pp", C.POINTER(C.c_float))]


def test_frameshift_oatrace_state_by_state(hs, gpu_ctx):
    """p7_OATrace_Frameshift through the shim against the oracle's p7_GOATrace_Frameshift (generic_optacc_frameshift.c:373-588)
    on planted frameshifted genes: every state, node, position and codon length of the trace."""
    model = ol.Model(PATH)
    rng = np.random.default_rng(5)
    import test_frameshift_gpu as tf
    envs = [w for w in tf.fs_windows(rng, model, 8, with_degenerate=False) if len(w) >= 60][:8]
    L_ = ol.lib()
    L_.bo_goatrace_fs.argtypes = [C.POINTER(ol.FsProfile), C.POINTER(ol.Gmx), C.POINTER(ol.Gmx), C.POINTER(OTrace)]
    L_.bo_trace_init.argtypes = [C.POINTER(OTrace)]; L_.bo_trace_free.argtypes = [C.POINTER(OTrace)]
    o2p = {0: 4, 1: 5, 2: 6, 3: 1, 4: 2, 5: 3, 6: 7, 7: 10, 8: 8, 9: 9}                  # BO_T_* -> p7T_*
    gm5 = model.fs(5)
    shifted = 0
    for w in envs:
        L = len(w)
        d = ol.dsq_from(w)
        out, null2 = np.zeros(3, np.float32), np.zeros(29, np.float32)
        cap = 4 * (L + model.M) + 64
        tst = C.create_string_buffer(cap)
        tk, ti, tc, tpp = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        n = hs.hs_fs_envelope(PATH.encode(), 0, ol.u8(d), L, fp(out), fp(null2), tst, ip(tk), ip(ti), ip(tc), fp(tpp), cap)
        assert n > 0
        L_.bo_fs_profile_reconfig_unihit(gm5, L // 3)
        g8, g3, oa = L_.bo_gmx_create(model.M, L + 1, L, 8), L_.bo_gmx_create(model.M, L + 1, L, 3), L_.bo_gmx_create(model.M, L + 1, L, 3)
        f, b, e = C.c_float(), C.c_float(), C.c_float()
        assert L_.bo_gforward_fs(ol.u8(d), L, gm5, g8, 0, C.byref(f)) == 0 and L_.bo_gbackward_fs(ol.u8(d), L, gm5, g3, C.byref(b)) == 0
        assert bits(np.float32(f.value)) == bits(out[0]) and bits(np.float32(b.value)) == bits(out[1])       # strict: bit for bit
        L_.bo_gdecoding_fs(gm5, g8, g3)
        L_.bo_goptacc_fs(gm5, g8, oa, C.byref(e))
        tr = OTrace(); L_.bo_trace_init(C.byref(tr))
        assert L_.bo_goatrace_fs(gm5, g8, oa, C.byref(tr)) == 0
        want = [(o2p[int(tr.st[z])], int(tr.k[z]), int(tr.i[z]), int(tr.c[z])) for z in range(tr.N)]
        L_.bo_trace_free(C.byref(tr))
        for g in (g8, g3, oa):
            L_.bo_gmx_free(g)
        st = [tst.raw[z] for z in range(n)]
        got = [(st[z], int(tk[z]), int(ti[z]), int(tc[z])) for z in range(n)]
        core = lambda t: [(a, k, i if a != T_D else 0, c) for a, k, i, c in t if a in (T_M, T_D, T_I)]     # p7_trace_Append: a D state carries no position
        assert core(got) == core(want)
        assert [a for a, _, _, _ in got] == [a for a, _, _, _ in want]
        shifted += sum(1 for a, _, _, c in got if a == T_M and c != 3)
    L_.bo_fs_profile_reconfig_multihit(gm5, 100)
    assert shifted >= 3                                                                   # the traces go through frameshifted codons


def test_frameshift_region_stochastic_traces(hs, gpu_ctx):
    model = ol.Model(PATH)
    hmm = ba.HMM(PATH)
    rng = np.random.default_rng(3)
    g = common.emit_from_model(rng, model, 2, flank=2, sharpen=2.0)
    nt = np.concatenate([common.revtranslate(rng, g[0], model.basic), rng.integers(0, 4, size=30).astype(np.uint8), common.revtranslate(rng, g[1], model.basic)])
    L = len(nt)
    d = ol.dsq_from(nt)
    sc = np.zeros(1, np.float32)
    nd, fi, li = np.zeros(40, np.int32), np.zeros(40, np.int32), np.zeros(40, np.int32)
    assert hs.hs_fs_region(PATH.encode(), 0, ol.u8(d), L, 42, 40, fp(sc), ip(nd), ip(fi), ip(li)) == 0
    om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5))
    bsc = np.zeros(1, np.float32)
    blk = ba.SeqBlock(gpu_ctx, [nt])
    gpu_ctx._check(ba.lib().bath_hip_fs5_forward_full(gpu_ctx._h, om5._h, blk._h, 100, fp(bsc), None, None), "fs5_forward_full")
    assert bits(sc[0]) == bits(bsc[0])
    assert nd.min() >= 1 and (nd >= 2).sum() >= 15 and fi.min() >= 1 and li.max() <= L
