"""Seeded synthetic inputs for the benchmark configurations of BASELINE.json.

C2: iid uniform ACGT windows (seed 42) with a planted set: a fraction of the windows carry a domain
sampled from the profile's match emissions and reverse-translated with uniformly chosen synonymous
codons, on either strand, so that the stages after MSV are exercised.  No file I/O, no network.
"""
import ctypes as C

import numpy as np

from . import gencode_basic


def hmm_match_emissions(hmm):
    M = hmm.M
    return np.ctypeslib.as_array(hmm._p.contents.mat, shape=((M + 1) * 20,)).reshape(M + 1, 20).copy()


def sample_domain(rng, mat, cum=None):
    """A (partial) pass through the match states, a few skipped nodes and short inserts."""
    M = mat.shape[0] - 1
    a = int(rng.integers(1, max(2, M // 3)))
    b = int(rng.integers(min(M, a + max(10, M // 3)), M + 1))
    ks = np.arange(a, b + 1)
    ks = ks[rng.random(len(ks)) > 0.04]
    u = rng.random(len(ks))
    cum = np.cumsum(mat[ks], axis=1)
    cum /= cum[:, -1:]
    return (u[:, None] > cum).sum(axis=1).clip(0, 19).astype(np.uint8)


def reverse_translate(rng, aa, basic):
    """Uniformly chosen synonymous codon per residue (vectorised)."""
    table = np.zeros((20, 6), dtype=np.int64)
    count = np.zeros(20, dtype=np.int64)
    for a in range(20):
        c = np.flatnonzero(basic == a)
        table[a, : len(c)] = c
        count[a] = len(c)
    aa = np.asarray(aa, dtype=np.int64)
    pick = (rng.random(len(aa)) * count[aa]).astype(np.int64)
    c = table[aa, pick]
    return np.stack([c >> 4, (c >> 2) & 3, c & 3], axis=1).reshape(-1).astype(np.uint8)


def frameshift_mutations(rng, nt):
    """per codon P(-1 nt) = P(+1 nt) = 0.01, P(-2) = P(+2) = 0.005, an in-frame stop (TAA) at 0.002."""
    out = []
    for j in range(0, len(nt) - 2, 3):
        c = list(nt[j:j + 3])
        r = rng.random()
        if r < 0.010:
            del c[int(rng.integers(0, 3))]
        elif r < 0.020:
            c.insert(int(rng.integers(0, 4)), int(rng.integers(0, 4)))
        elif r < 0.025:
            c = c[:1]
        elif r < 0.030:
            c = c + [int(rng.integers(0, 4)), int(rng.integers(0, 4))]
        elif r < 0.032:
            c = [3, 0, 0]
        out.extend(c)
    return np.asarray(out, dtype=np.uint8)


def dna_windows(n_windows, length, seed, hmm=None, planted_frac=0.01, ncbi_table=1, frameshift=False):
    """Returns (flat uint8 codes [n_windows*length], int64 offsets[n_windows+1], planted window indices).
    frameshift: the planted domains carry indels and in-frame stops (BASELINE configs[2])."""
    rng = np.random.default_rng(seed)
    flat = rng.integers(0, 4, size=(n_windows, length), dtype=np.uint8)
    planted = np.zeros(0, dtype=np.int64)
    if hmm is not None and planted_frac > 0 and n_windows > 0:
        basic = gencode_basic(ncbi_table)
        mat = hmm_match_emissions(hmm)
        n_pl = max(1, int(round(n_windows * planted_frac)))
        planted = np.sort(rng.choice(n_windows, size=n_pl, replace=False))
        for w in planted:
            nt = reverse_translate(rng, sample_domain(rng, mat), basic)[: length - 2]
            if frameshift:
                nt = frameshift_mutations(rng, nt)[: length - 2]
            pos = int(rng.integers(0, length - len(nt) + 1))
            if rng.random() < 0.5:
                nt = (3 - nt[::-1]).astype(np.uint8)
            flat[w, pos:pos + len(nt)] = nt
    offsets = np.arange(n_windows + 1, dtype=np.int64) * length
    return flat.reshape(-1), offsets, planted


BG = np.array([0.0787945, 0.0151600, 0.0535222, 0.0668298, 0.0397062, 0.0695071, 0.0229198, 0.0590092,
               0.0594422, 0.0963728, 0.0237718, 0.0414386, 0.0482904, 0.0395639, 0.0540978, 0.0683364,
               0.0540687, 0.0673417, 0.0114135, 0.0304133])


def write_synthetic_bhmm(path, M, seed=1, name="synth", evparam_from=None):
    """A synthetic BATH3/f model in the spirit of p7_hmm_Sample (p7_hmm.c): Dirichlet-ish match emissions, sampled
    transitions; E-value parameters copied from a real model (BASELINE configs[4]: 'synthetic 1024-state HMM')."""
    rng = np.random.default_rng(seed)
    ev = evparam_from if evparam_from is not None else [-10.3292, 0.71002, -11.3545, 0.71002, -3.7358, 0.71002, -4.1064, -3.2990]
    ins = BG / BG.sum()

    def line(vals):
        return "  ".join("%.5f" % (-np.log(max(v, 1e-30))) if v > 0 else "      *" for v in vals)

    out = ["BATH3/f", "NAME  %s" % name, "LENG  %d" % M, "MAXL  %d" % int(M * 1.6 + 50), "ALPH  amino", "RF    no", "MM    no",
           "CONS  yes", "CS    no", "MAP   no", "NSEQ  1", "EFFN  1.000000",
           "STATS LOCAL MSV       %9.4f  %7.5f" % (ev[0], ev[1]), "STATS LOCAL VITERBI   %9.4f  %7.5f" % (ev[2], ev[3]),
           "STATS LOCAL FORWARD   %9.4f  %7.5f" % (ev[4], ev[5]), "STATS LOCAL FS3 FORWARD  %8.4f  %7.5f" % (ev[6], ev[5]),
           "STATS LOCAL FS5 FORWARD  %8.4f  %7.5f" % (ev[7], ev[5]), "FRAMESHIFT PROB    0.0100", "CODON TABLE  1",
           "HMM          " + "        ".join("ACDEFGHIKLMNPQRSTVWY"), "            m->m     m->i     m->d     i->m     i->i     d->m     d->d"]
    mats, trans = [], []
    for k in range(M + 1):
        e = rng.dirichlet(0.3 * np.ones(20)) * 0.7 + 0.3 * ins
        mats.append(e / e.sum())
        tm = rng.dirichlet([30.0, 1.0, 1.0]); ti = rng.dirichlet([2.0, 1.5]); td = rng.dirichlet([1.5, 1.0])
        trans.append([tm[0], tm[1], tm[2], ti[0], ti[1], td[0], td[1]])
    trans[0][5], trans[0][6] = 1.0, 0.0
    trans[M] = [trans[M][0] / (trans[M][0] + trans[M][1]), trans[M][1] / (trans[M][0] + trans[M][1]), 0.0, trans[M][3], trans[M][4], 1.0, 0.0]
    compo = np.mean(mats[1:], axis=0)
    out.append("  COMPO   " + line(compo))
    out.append("          " + line(ins))
    out.append("          " + line(trans[0]))
    for k in range(1, M + 1):
        cons = "ACDEFGHIKLMNPQRSTVWY"[int(np.argmax(mats[k]))].lower()
        out.append("%7d   %s %6d %s - -" % (k, line(mats[k]), k, cons))
        out.append("          " + line(ins))
        out.append("          " + line(trans[k]))
    out.append("//")
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")
    return path


def genome(n_nt, seed, hmms=(), genes_per_model=0, ncbi_table=1, frameshift=False):
    """One iid ACGT target of <n_nt> nucleotides with <genes_per_model> genes sampled from each model's match emissions planted
    on either strand (BASELINE configs[3] / [4]).  Returns (codes, [(model index, position, length)])."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=n_nt, dtype=np.uint8)
    basic = gencode_basic(ncbi_table)
    planted = []
    for q, hmm in enumerate(hmms):
        mat = hmm_match_emissions(hmm)
        for j in range(genes_per_model):
            nt = reverse_translate(rng, sample_domain(rng, mat), basic)
            if frameshift:
                nt = frameshift_mutations(rng, nt)
            if rng.random() < 0.5:
                nt = (3 - nt[::-1]).astype(np.uint8)
            p = int(rng.integers(0, n_nt - len(nt)))
            g[p:p + len(nt)] = nt
            planted.append((q, p, len(nt)))
    return g, planted
