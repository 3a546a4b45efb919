"""Six-frame translation + ORF finding on the GPU (bath_hip_translate_orfs) against the oracle's restatement of
esl_gencode_ProcessStart/Piece/End (oracle/translate.c), ORF by ORF: coordinates and residues identical.

Covers what the tile decomposition has to get right: windows shorter than one tile, windows of many tiles with ORFs
crossing tile edges (long stop-free runs), lengths of every residue class mod 3 and mod 12, windows under 15 nt,
degenerate nucleotides, alternative genetic codes and other minimum lengths."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def oracle_orfs(windows, ct=1, minlen=20):
    """[(window, strand, frame, start, end, residues)] sorted like the GPU list."""
    L_ = ol.lib()
    basic = np.zeros(64, np.uint8)
    assert L_.bo_gencode_basic(ct, ol.u8(basic)) == 0
    out = []
    blk = ol.OrfBlock(); L_.bo_orfblock_init(C.byref(blk))
    for w, codes in enumerate(windows):
        n = len(codes)
        if n < 15:                                  # bathsearch.c:1066
            continue
        d = ol.dsq_from(codes)
        rc = np.zeros(n + 2, np.uint8)
        L_.bo_revcomp(ol.u8(d), n, ol.u8(rc))
        for strand, dsq in ((0, d), (1, rc)):
            L_.bo_orfblock_reuse(C.byref(blk))
            L_.bo_translate_orfs(ol.u8(dsq), n, ol.u8(basic), minlen, C.byref(blk))
            if blk.count == 0:
                continue
            aa = np.ctypeslib.as_array(blk.aa, shape=(int(blk.aa_n),))
            for i in range(blk.count):
                o = blk.orf[i]
                out.append((w, strand, o.frame, o.start, o.end, aa[o.off + 1:o.off + 1 + o.n].copy()))
    L_.bo_orfblock_free(C.byref(blk))
    out.sort(key=lambda r: r[:4])
    return out


def check(gpu_ctx, windows, ct=1, minlen=20):
    import bath_amd as ba
    dna = ba.SeqBlock(gpu_ctx, [np.asarray(w, np.uint8) for w in windows])
    got = ba.translate_orfs(gpu_ctx, dna, ct, minlen)
    want = oracle_orfs(windows, ct, minlen)
    assert len(got) == len(want), (len(got), len(want))
    for g, o in zip(got, want):
        assert tuple(g[:5]) == tuple(o[:5]), (g[:5], o[:5])
        assert np.array_equal(g[5], o[5]), (g[:5],)
    return len(got)


def rand_dna(rng, n, p_degen=0.0, stop_poor=False):
    if stop_poor:       # few T and A: long stop-free runs, ORFs spanning many tiles
        x = rng.choice(4, size=n, p=[0.04, 0.46, 0.46, 0.04]).astype(np.uint8)
    else:
        x = rng.integers(0, 4, size=n, dtype=np.uint8)
    if p_degen > 0:
        m = rng.random(n) < p_degen
        x[m] = rng.choice([5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], size=int(m.sum())).astype(np.uint8)
    return x


def test_lengths_around_tiles(gpu_ctx):
    rng = np.random.default_rng(5)
    lens = list(range(0, 40)) + [383, 384, 385, 386, 387, 395, 396, 397, 767, 768, 769, 770, 1000, 1151, 1152, 1153, 1154, 1155]
    n = check(gpu_ctx, [rand_dna(rng, L) for L in lens for _ in range(3)])
    assert n > 50


def test_long_windows_cross_tile_orfs(gpu_ctx):
    rng = np.random.default_rng(6)
    wins = [rand_dna(rng, L, stop_poor=True) for L in (5000, 12345, 40001, 100002)] + [rand_dna(rng, 30000)]
    n = check(gpu_ctx, wins)
    assert n > 100


def test_no_stop_at_all(gpu_ctx):
    wins = [np.full(L, 1, np.uint8) for L in (15, 59, 60, 62, 383, 384, 385, 1536, 1537, 5000)]    # poly-C: one ORF per frame
    n = check(gpu_ctx, wins)
    assert n >= 6 * 7


def test_degenerate_nucleotides(gpu_ctx):
    rng = np.random.default_rng(7)
    check(gpu_ctx, [rand_dna(rng, int(L), p_degen=0.02) for L in rng.integers(15, 3000, size=60)])
    check(gpu_ctx, [rand_dna(rng, 2000, p_degen=0.3, stop_poor=True)])


@pytest.mark.parametrize("ct", [4, 11, 6])
def test_other_genetic_codes(gpu_ctx, ct):
    rng = np.random.default_rng(8 + ct)
    check(gpu_ctx, [rand_dna(rng, int(L)) for L in rng.integers(15, 2500, size=40)], ct=ct)


@pytest.mark.parametrize("minlen", [1, 2, 5, 64, 200])
def test_other_minimum_lengths(gpu_ctx, minlen):
    rng = np.random.default_rng(20 + minlen)
    check(gpu_ctx, [rand_dna(rng, int(L), stop_poor=(minlen >= 64)) for L in rng.integers(15, 4000, size=30)], minlen=minlen)


def test_many_short_windows(gpu_ctx):
    rng = np.random.default_rng(9)
    n = check(gpu_ctx, [rand_dna(rng, 1000) for _ in range(400)])
    assert n > 10000


@pytest.mark.parametrize("wave", ["1", "0"])
def test_both_stitch_kernels_on_every_input(wave):
    """orf_stitch_wave_kernel (a wave per (window, frame): the default for blocks of long windows) forced onto every input of this
    file, short windows and stop-free windows included, and the lane-per-stream kernel forced onto the long windows: fresh
    processes (the switch is read once), same ORF lists."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, BATH_HIP_STITCH_WAVE=wave)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(here, "test_translate_gpu.py"), "-k", "not stitch_kernels"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=os.path.dirname(here))
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
