"""The option state of the pipeline (p7_pipeline_Create_BATH, p7_pipeline.c:94-234; bathsearch.c:718-719, :831-833, :868-881)
without a GPU: the oracle's switches against what each one means, and the host-side pieces of the library that carry them
(bath_pipeline_params_default, bath_gencode_initiators, bath_tophits_set_score_thresholds, bath_search_space_residues).
The GPU path is held against the oracle switch by switch in tests/test_options_gpu.py."""
import ctypes as C

import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol
from test_tophits_cpu import mk

CAUDAL = ol.GOLDEN + "/Caudal_act.bhmm"
TABLES = (1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 13, 14, 16, 21, 22, 23, 24, 25)


def planted(model, seed=3, n=12):
    rng = np.random.default_rng(seed)
    wins = []
    for i, aa in enumerate(common.emit_from_model(rng, model, n, flank=5, sharpen=2.0)):
        nt = list(common.revtranslate(rng, [10] + list(aa), model.basic))
        if i % 3 == 0:
            del nt[len(nt) // 2]
        w = np.concatenate([rng.integers(0, 4, size=int(rng.integers(3, 150))).astype(np.uint8), np.array(nt, dtype=np.uint8),
                            rng.integers(0, 4, size=int(rng.integers(0, 150))).astype(np.uint8)])
        wins.append((3 - w[::-1]).astype(np.uint8) if i % 2 else w)
    return wins + common.random_dna(rng, 6, 600)


def test_defaults():
    p = ba.PipelineParams()
    ba.lib().bath_pipeline_params_default(p, 0)
    assert (p.do_null2, p.std_pipe, p.strands, p.initiator, p.inc_by_E, p.seed, p.T) == (1, 1, 0, 0, 1, 42, 0.0)
    pli = ol.Pipeline(); ol.lib().bo_pipeline_init(C.byref(pli), 0)
    assert (pli.do_null2, pli.std_pipe, pli.strands, pli.initiator, pli.inc_by_E, pli.seed, pli.T) == (1, 1, 0, 0, 1, 42, 0.0)


@pytest.mark.parametrize("ct", TABLES)
def test_initiator_tables_product_equals_oracle(ct):
    for mode in (0, 1, 2):
        a = np.zeros(64, np.uint8); b = np.zeros(64, np.uint8)
        assert ba.lib().bath_gencode_initiators(ct, mode, a.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
        assert ol.lib().bo_gencode_initiators(ct, mode, ol.u8(b)) == 0
        assert np.array_equal(a, b)
        atg = 16 * 0 + 4 * 3 + 2
        if mode == 0:
            assert a.all()
        elif mode == 2:
            assert a.sum() == 1 and a[atg]
        else:
            basic = np.zeros(64, np.uint8); ol.lib().bo_gencode_basic(ct, ol.u8(basic))
            assert a[atg] and 1 <= a.sum() <= 8
            assert not any(a[c] and basic[c] == 27 for c in range(64))      # no table lists a stop as a start


def test_initiator_orfs_are_suffixes_of_the_any_codon_orfs():
    """esl_gencode_ProcessPiece with initiators: per stop-free run the ORF from its first initiation codon on, that codon as M."""
    L_ = ol.lib()
    rng = np.random.default_rng(4)
    basic = np.zeros(64, np.uint8); L_.bo_gencode_basic(1, ol.u8(basic))
    for mode in (1, 2):
        is_init = np.zeros(64, np.uint8); L_.bo_gencode_initiators(1, mode, ol.u8(is_init))
        seq = rng.integers(0, 4, size=6000).astype(np.uint8)
        d = ol.dsq_from(seq)
        a = ol.OrfBlock(); L_.bo_orfblock_init(C.byref(a)); b = ol.OrfBlock(); L_.bo_orfblock_init(C.byref(b))
        L_.bo_translate_orfs(ol.u8(d), len(seq), ol.u8(basic), 1, C.byref(a))
        L_.bo_translate_orfs_init(ol.u8(d), len(seq), ol.u8(basic), ol.u8(is_init), 1, 5, C.byref(b))
        runs = {(a.orf[i].frame, a.orf[i].end): a.orf[i] for i in range(a.count)}
        assert b.count > 10
        aa_b = np.ctypeslib.as_array(b.aa, shape=(int(b.aa_n),)); aa_a = np.ctypeslib.as_array(a.aa, shape=(int(a.aa_n),))
        for i in range(b.count):
            o = b.orf[i]
            r = runs[(o.frame, o.end)]                                   # same closing stop
            assert o.start >= r.start and (o.start - r.start) % 3 == 0 and o.n >= 5
            codon = seq[o.start - 1:o.start + 2]
            assert is_init[16 * codon[0] + 4 * codon[1] + codon[2]]
            for p in range(r.start, o.start, 3):                         # nothing before it in the run initiates
                c = seq[p - 1:p + 2]
                assert not is_init[16 * c[0] + 4 * c[1] + c[2]]
            ra = aa_a[r.off + 1:r.off + 1 + r.n]; rb = aa_b[o.off + 1:o.off + 1 + o.n]
            assert rb[0] == 10 and np.array_equal(rb[1:], ra[len(ra) - len(rb) + 1:])
        L_.bo_orfblock_free(C.byref(a)); L_.bo_orfblock_free(C.byref(b))


def test_oracle_one_strand_is_that_strands_share_of_both():
    model = ol.Model(CAUDAL, 0)
    wins = planted(model)
    pli2, odm2, per2, _ = model.run_pipeline_hits(wins)
    key = lambda w, d: (w, d.ienv, d.jenv, d.iali, d.jali, d.ihmm, d.jhmm)
    both = {key(w, o) for w, (a, b) in enumerate(per2) for o in odm2[a:b]}
    got = {}
    for strands in (1, 2):
        pli, odm, per, _ = model.run_pipeline_hits(wins, opts={"strands": strands})
        assert pli.nres * 2 == pli2.nres
        got[strands] = {key(w, o) for w, (a, b) in enumerate(per) for o in odm[a:b]}
        assert got[strands] and all((k[1] < k[2]) == (strands == 1) for k in got[strands])
    assert got[1] | got[2] == both and not (got[1] & got[2])


def test_oracle_nonull2_fsonly_and_score_threshold():
    model = ol.Model(CAUDAL, 0)
    wins = planted(model)
    pli, ofw, per_w, odm, per_d, _ = model.run_pipeline_fsdom(wins)
    assert any(o.dombias > 0 for o in odm) and {w.branch for w in ofw} == {1, 2}
    _, ofw0, _, odm0, _, _ = model.run_pipeline_fsdom(wins, opts={"do_null2": 0})
    assert len(odm0) == len(odm) and all(o.dombias == 0.0 for o in odm0)
    assert all(a.bitscore >= b.bitscore for a, b in zip(odm0, odm))
    _, ofw1, _, odm1, _, _ = model.run_pipeline_fsdom(wins, opts={"std_pipe": 0})
    assert {w.branch for w in ofw1} <= {0, 1} and all(w.P_tot == 1.0 for w in ofw1)
    assert sum(w.branch == 1 for w in ofw1) >= sum(w.branch == 1 for w in ofw)
    T = float(np.median([o.bitscore for o in odm]))
    _, _, _, odmT, _, _ = model.run_pipeline_fsdom(wins, opts={"inc_by_E": 0, "T": T})
    assert [o.reported for o in odmT] == [1 if o.bitscore >= T else 0 for o in odmT] and 0 < sum(o.reported for o in odmT) < len(odmT)


def test_tophits_by_score_and_search_space():
    doms = [mk(0, 100, 400, 1, 100, -40.0, score=60.0), mk(0, 1000, 1300, 1, 100, -2.0, score=25.0), mk(1, 100, 400, 1, 100, -30.0, score=12.0)]
    names, lens = ["a", "b"], [10000, 10000]
    th = ba.TopHits(); th.add(doms, names, lens)
    th.finalize(nres=300, max_length=100, E=1e-5)                           # by E: log(300 / 300) = 0 correction
    assert th.reported() == 2
    th = ba.TopHits(); th.add(doms, names, lens)
    th.set_score_thresholds(by_E=False, T=20.0)                             # -T 20 (p7_pli_TargetReportable)
    th.finalize(nres=300, max_length=100, E=1e-5)
    flags = {d.bitscore: fl for d, _, fl in th.hits()}
    assert th.reported() == 2 and flags[60.0] & 1 and flags[25.0] & 1 and not flags[12.0] & 1
    assert flags[60.0] & 2 and not flags[25.0] & 2                          # inclusion still by E (incE = 0.01)
    th = ba.TopHits(); th.add(doms, names, lens)
    th.set_score_thresholds(by_E=True, T=0.0, inc_by_E=False, incT=20.0)    # --incT 20 (p7_pli_TargetIncludable)
    th.finalize(nres=300, max_length=100, E=10.0)
    flags = {d.bitscore: fl for d, _, fl in th.hits()}
    assert flags[60.0] & 2 and flags[25.0] & 2 and not flags[12.0] & 2
    f = ba.lib().bath_search_space_residues
    assert f(0, 0.0, 0, 123456) == 123456                                    # no -Z: the residues searched
    assert f(1, 2.5, ba.STRAND_BOTH, 123456) == 5_000_000                    # -Z 2.5: 2.5e6 per strand searched (bathsearch.c:870-874)
    assert f(1, 2.5, ba.STRAND_TOPONLY, 123456) == 2_500_000
