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
