"""CPU tier: self-consistency of the frameshift oracle, the properties the reference's own unit test checks
(utest_forward_fs, generic_fwdback_frameshift.c:2304-2435: |Fwd - Bwd| <= 0.001 with exact log-sums; scores
of model-emitted DNA above random DNA) plus the quirk documented in DESIGN.md (5-nt codon ring aliasing)."""
import ctypes as C

import numpy as np
import pytest

import common
import oracle_lib as ol


@pytest.fixture(scope="module")
def model():
    return ol.Model(ol.GOLDEN + "/2OG-FeII_Oxy_3.bhmm")


def shifted_dna(rng, model, n, sharpen=1.0):
    out = []
    for aa in common.emit_from_model(rng, model, n, flank=5, sharpen=sharpen):
        nt = list(common.revtranslate(rng, aa, model.basic))
        if len(nt) > 60:
            del nt[30]
            nt.insert(50, 1)
        out.append(np.array(nt, dtype=np.uint8))
    return out


def run_fs5(model, w, compat, exact):
    L_ = ol.lib()
    L_.bo_flogsum_set_exact(1 if exact else 0)
    gm5 = model.fs(5)
    L = len(w)
    L_.bo_fs_profile_reconfig_unihit(gm5, L // 3)
    g8 = L_.bo_gmx_create(model.M, L + 1, L, 8)
    g3 = L_.bo_gmx_create(model.M, L + 1, L, 3)
    f, b = C.c_float(), C.c_float()
    d = ol.dsq_from(w)
    assert L_.bo_gforward_fs(ol.u8(d), L, gm5, g8, compat, C.byref(f)) == 0
    assert L_.bo_gbackward_fs(ol.u8(d), L, gm5, g3, C.byref(b)) == 0
    L_.bo_gdecoding_fs(gm5, g8, g3)
    pp = np.ctypeslib.as_array(g8.contents.dp, shape=(L + 1, model.M + 1, 8)).copy()
    xm = np.ctypeslib.as_array(g8.contents.xmx, shape=(L + 1, 5)).copy()
    L_.bo_gmx_free(g8); L_.bo_gmx_free(g3)
    L_.bo_fs_profile_reconfig_multihit(gm5, 100)
    L_.bo_flogsum_set_exact(0)
    return f.value, b.value, pp, xm


def test_forward_equals_backward_exact_logsum(model):
    rng = np.random.default_rng(9)
    for w in shifted_dna(rng, model, 6):
        f, b, pp, xm = run_fs5(model, w, compat=0, exact=True)
        assert abs(f - b) <= 1e-3, (f, b)
        # posterior rows are normalised over the emitting states (generic_decoding_frameshift.c:131-152)
        rows = pp[1:, :, 2].sum(1) + pp[1:, :, 1].sum(1) + xm[1:, 1] + xm[1:, 2] + xm[1:, 4]
        assert np.allclose(rows, 1.0, atol=1e-4)
        # per-codon-length match posteriors add up to the total (hmmer.h:611-619)
        assert np.abs(pp[5:, :, 3:8].sum(2) - pp[5:, :, 2]).max() < 1e-4


def test_generic_five_nt_ring_quirk_is_reproducible(model):
    """generic_fwdback_frameshift.c:324 reads slot (i-5)%5 == i%5; with it Forward no longer equals Backward."""
    rng = np.random.default_rng(10)
    gaps = []
    for w in shifted_dna(rng, model, 6):
        f1, b1, _, _ = run_fs5(model, w, compat=1, exact=True)
        f0, b0, _, _ = run_fs5(model, w, compat=0, exact=True)
        assert b0 == b1 and abs(f0 - b0) <= 1e-3
        gaps.append(abs(f1 - b1))
    assert max(gaps) > 1e-3


def test_fs3_parsers_and_scores_beat_random(model):
    L_ = ol.lib()
    rng = np.random.default_rng(11)
    gm3 = model.fs(3)
    f, b = C.c_float(), C.c_float()

    def both(w):
        L = len(w)
        d = ol.dsq_from(w)
        L_.bo_fs_profile_reconfig_length(gm3, L // 3)
        gx = L_.bo_gmx_create(model.M, L + 1, L, 3)
        assert L_.bo_gforward_parser_fs3(ol.u8(d), L, gm3, gx, C.byref(f)) == 0
        assert L_.bo_gbackward_parser_fs3(ol.u8(d), L, gm3, gx, C.byref(b)) == 0
        L_.bo_gmx_free(gx)
        return f.value, b.value

    L_.bo_flogsum_set_exact(1)
    hom = [both(w) for w in shifted_dna(rng, model, 5, sharpen=3.0)]
    rnd = [both(w) for w in common.random_dna(rng, 5, 270)]
    short = [both(rng.integers(0, 4, size=n).astype(np.uint8)) for n in (15, 16, 17, 21)]
    L_.bo_flogsum_set_exact(0)
    for fv, bv in hom + rnd + short:
        assert abs(fv - bv) <= 1e-3, (fv, bv)
    assert np.mean([h[0] for h in hom]) > np.mean([r[0] for r in rnd]) + 10.0
