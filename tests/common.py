"""Shared helpers for the test-suite: seeded synthetic inputs (the reference's unit tests use sampled
models and sequences with fixed seeds, e.g. msvfilter.c:706-727, never files)."""
import ctypes as C

import numpy as np

import oracle_lib as ol

BG = np.array([0.0787945, 0.0151600, 0.0535222, 0.0668298, 0.0397062, 0.0695071, 0.0229198, 0.0590092,
               0.0594422, 0.0963728, 0.0237718, 0.0414386, 0.0482904, 0.0395639, 0.0540978, 0.0683364,
               0.0540687, 0.0673417, 0.0114135, 0.0304133])


def random_aa(rng, n, L_lo=20, L_hi=400, with_degenerate=True):
    """iid background amino sequences (esl_rsq_xfIID analogue), a few with degenerate residues."""
    seqs = []
    p = BG / BG.sum()
    for _ in range(n):
        L = int(rng.integers(L_lo, L_hi + 1))
        s = rng.choice(20, size=L, p=p).astype(np.uint8)
        if with_degenerate and rng.random() < 0.1 and L > 0:
            idx = rng.integers(0, L, size=max(1, L // 20))
            s[idx] = rng.choice([21, 22, 23, 24, 25, 26], size=len(idx))
        seqs.append(s)
    return seqs


def emit_from_model(rng, model, n, flank=30, sharpen=1.0):
    """Sequences carrying a (partial) pass through the model's match states with flanks: homologs that
    exercise the later cascade stages and the overflow/J-state branches (p7_ProfileEmit analogue)."""
    h = model.hmm.contents
    M = h.M
    mat = np.ctypeslib.as_array(h.mat, shape=((M + 1) * 20,)).reshape(M + 1, 20)
    p = BG / BG.sum()
    out = []
    for _ in range(n):
        a = int(rng.integers(1, max(2, M // 2)))
        b = int(rng.integers(min(M, a + 10), M + 1))
        core = []
        for k in range(a, b + 1):
            if rng.random() < 0.05:
                continue
            q = mat[k].astype(np.float64) ** sharpen
            q /= q.sum()
            core.append(rng.choice(20, p=q))
            if rng.random() < 0.03:
                core.extend(rng.choice(20, size=int(rng.integers(1, 4)), p=p))
        reps = 1 if rng.random() < 0.7 else 2
        body = []
        for r in range(reps):
            body.extend(core)
            if r + 1 < reps:
                body.extend(rng.choice(20, size=int(rng.integers(5, 40)), p=p))
        left = rng.choice(20, size=int(rng.integers(0, flank)), p=p)
        right = rng.choice(20, size=int(rng.integers(0, flank)), p=p)
        out.append(np.concatenate([left, np.array(body, dtype=np.int64), right]).astype(np.uint8))
    return out


def oracle_scores(model, seqs, fn_name):
    """Run an oracle filter per sequence after p7_oprofile_ReconfigLength(om, L) (p7_pipeline.c:1644)."""
    L = ol.lib()
    fn = getattr(L, fn_name)
    sc = np.zeros(len(seqs), dtype=np.float32)
    st = np.zeros(len(seqs), dtype=np.int32)
    out = C.c_float(0)
    for i, s in enumerate(seqs):
        d = ol.dsq_from(s)
        L.bo_oprofile_reconfig_length(model.om, len(s))
        out.value = 0.0
        if fn_name == "bo_forward_parser":
            st[i] = fn(ol.u8(d), len(s), model.om, None, C.byref(out))
        else:
            st[i] = fn(ol.u8(d), len(s), model.om, C.byref(out))
        sc[i] = out.value
    return sc, st


def oracle_bias(model, seqs):
    L = ol.lib()
    nullsc = np.zeros(len(seqs), dtype=np.float32)
    fsc = np.zeros(len(seqs), dtype=np.float32)
    L.bo_bg_setfilter(C.byref(model.bg), model.M, model.om.contents.compo)
    for i, s in enumerate(seqs):
        d = ol.dsq_from(s)
        L.bo_bg_setlength(C.byref(model.bg), len(s))
        nullsc[i] = L.bo_bg_nullone(C.byref(model.bg), len(s))
        fsc[i] = L.bo_bg_filterscore(C.byref(model.bg), ol.u8(d), len(s))
    return nullsc, fsc


def random_dna(rng, n, L=1000, degenerate_frac=0.0):
    seqs = []
    for _ in range(n):
        s = rng.integers(0, 4, size=L).astype(np.uint8)
        if degenerate_frac > 0:
            m = rng.random(L) < degenerate_frac
            s[m] = rng.choice([5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], size=int(m.sum()))
        seqs.append(s)
    return seqs


def revtranslate(rng, aa, basic):
    """Uniformly chosen synonymous codons (p7_codontable_GetCodon analogue); X and friends -> random codon."""
    by_aa = {}
    for c in range(64):
        by_aa.setdefault(int(basic[c]), []).append(c)
    out = []
    for a in aa:
        cands = by_aa.get(int(a))
        if not cands:
            cands = [c for c in range(64) if basic[c] < 20]
        c = cands[int(rng.integers(0, len(cands)))]
        out.extend([c >> 4, (c >> 2) & 3, c & 3])
    return np.array(out, dtype=np.uint8)


def write_synthetic_bhmm(path, M, seed=1, name="synth", evparam_from=None):
    """A synthetic BATH3/f model (bath_amd.synth: the bench's configs[4] leg uses the same generator)."""
    from bath_amd import synth
    return synth.write_synthetic_bhmm(path, M, seed=seed, name=name, evparam_from=evparam_from)
