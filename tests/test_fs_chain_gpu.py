"""The strict chain kernels (bath_fs_chain.hip) in the launch shapes the bench's --fs pass uses -- hundreds to thousands of DNA
windows of mixed length -- against the oracle's generic_fwdback_frameshift.c restatement, bit for bit.

The small batches of tests/test_frameshift_gpu.py (8-40 windows) reduce every chain block to one window per block: chain_waves()
cuts W to 1, chain_batches() makes batches of one and fs3_fwd_chain_half_kernel is never chosen.  Here:
  * ~700 windows, 15..900 nt: blocks of W > 1 windows, a chain wave carrying 2W rows, batches by length with cnt < W in the last
    block, windows shorter than the batch's Lmax (waves that keep only the barriers), the special-state hand-over between lanes;
  * the half-wave Forward kernel (32 windows per block, all 64 chain lanes carrying a row), forced with BATH_HIP_FS_HALFWAVE=1 in
    a fresh process (the switches are read once into statics), with the default batches and with uniform batches of 3;
  * the regions' multihit 5-codon Forward (fs5_fwd_chain_kernel) with W > 1.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import bath_amd as ba
import common
import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def identical(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return a.shape == b.shape and bool(np.all(a.view(np.uint32) == b.view(np.uint32)))


def mixed_windows(rng, model, n_model, n_random, lo=15, hi=900):
    """Frameshifted model emissions (cut to random lengths) and random DNA, lengths spread over lo..hi, in random order."""
    from test_frameshift_gpu import fs_windows
    wins = []
    base = fs_windows(rng, model, n_model, with_degenerate=True)
    for w in base:
        L = int(rng.integers(lo, max(lo + 1, min(hi, len(w)) + 1)))
        s = int(rng.integers(0, len(w) - L + 1)) if len(w) > L else 0
        wins.append(w[s:s + L].copy())
    for _ in range(n_random):
        wins.append(rng.integers(0, 4, size=int(rng.integers(lo, hi + 1))).astype(np.uint8))
    order = rng.permutation(len(wins))
    return [wins[i] for i in order]


def oracle_fs3(model, wins, backward):
    from test_frameshift_gpu import oracle_fs3 as o
    return o(model, wins, backward=backward)


def check_fs3(ctx, model, om3, wins, what):
    blk = ba.SeqBlock(ctx, wins)
    for backward, fn in ((False, ba.FS3ForwardParser), (True, ba.FS3BackwardParser)):
        sc, xm = fn(ctx, om3, blk, logsum=ba.LOGSUM_TABLE_SERIAL, want_xmx=True)
        osc, oxm = oracle_fs3(model, wins, backward)
        bad = [i for i in range(len(wins)) if not identical(sc[i], osc[i])]
        assert not bad, (what, backward, "scores differ", bad[:8], [len(wins[i]) for i in bad[:8]])
        bad = [i for i, (g, o) in enumerate(zip(xm, oxm)) if not identical(g, o)]
        assert not bad, (what, backward, "special-state rows differ", bad[:8], [len(wins[i]) for i in bad[:8]])


@pytest.fixture(scope="module")
def setup(gpu_ctx):
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    model = ol.Model(path)
    hmm = ba.HMM(path)
    om3 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 3))
    om5 = ba.FSOProfile(gpu_ctx, ba.FSProfile(hmm, 5))
    return gpu_ctx, model, om3, om5


@pytest.mark.parametrize("n_model,n_random,hi", [(260, 440, 900), (500, 1750, 380)])
def test_fs3_chain_blocks_of_many_windows_are_bit_identical(setup, n_model, n_random, hi):
    """Full-wave chain kernels with W = 4 (700 windows) and W = 16 (2250 windows: more than 8 per CU) windows per block, batches by
    length (Forward: uniform blocks; Backward: chain_batches)."""
    ctx, model, om3, om5 = setup
    rng = np.random.default_rng(2024 + hi)
    wins = mixed_windows(rng, model, n_model, n_random, hi=hi)
    assert len(wins) >= n_model + n_random
    check_fs3(ctx, model, om3, wins, "full-wave, mixed lengths")


def test_fs3_chain_uniform_lengths_and_short_tail(setup):
    """Many windows of ONE length (every batch full, all rows of a chain wave end together) plus a handful of very short ones
    (L = 15..20: row types 'no codon fits' / 'fewer than three codon lengths fit' inside a block of long windows)."""
    ctx, model, om3, om5 = setup
    rng = np.random.default_rng(7)
    wins = [rng.integers(0, 4, size=301).astype(np.uint8) for _ in range(300)]
    wins += [rng.integers(0, 4, size=L).astype(np.uint8) for L in (15, 16, 17, 18, 19, 20, 15, 16)]
    check_fs3(ctx, model, om3, wins, "uniform + short tail")


def test_fs5_region_forward_chain_many_regions(setup):
    """fs5_fwd_chain_kernel with several regions per block: score, the whole matrix and the special-state rows of every region."""
    ctx, model, om3, om5 = setup
    rng = np.random.default_rng(11)
    env = mixed_windows(rng, model, 300, 420, lo=15, hi=400)          # > 2 x CUs regions: W = 2 and more
    eb = ba.SeqBlock(ctx, env)
    M = model.M
    foff = np.zeros(len(env) + 1, np.int64); np.cumsum([(len(w) + 1) * (M + 1) * 8 for w in env], out=foff[1:])
    xoff = np.zeros(len(env) + 1, np.int64); np.cumsum([(len(w) + 1) * 5 for w in env], out=xoff[1:])
    sc = np.zeros(len(env), np.float32); fwd = np.zeros(int(foff[-1]), np.float32); xmx = np.zeros(int(xoff[-1]), np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ctx.set_fs_strict(True)
    ctx._check(ba.lib().bath_hip_fs5_forward_full(ctx._h, om5._h, eb._h, 100, fp(sc), fp(fwd), fp(xmx)), "fs5_forward_full")
    L_ = ol.lib()
    gm5 = model.fs(5)
    L_.bo_fs_profile_reconfig_multihit(gm5, 100)
    f = C.c_float()
    for e, w in enumerate(env):
        L = len(w)
        g8 = L_.bo_gmx_create(M, L + 1, L, 8)
        assert L_.bo_gforward_fs(ol.u8(ol.dsq_from(w)), L, gm5, g8, 0, C.byref(f)) == 0
        dp = np.ctypeslib.as_array(g8.contents.dp, shape=(L + 1, M + 1, 8))
        ox = np.ctypeslib.as_array(g8.contents.xmx, shape=(L + 1, 5))
        ok = identical(np.float32(sc[e]), np.float32(f.value)) and identical(xmx[xoff[e]:xoff[e + 1]].reshape(L + 1, 5), ox) and \
            identical(fwd[foff[e]:foff[e + 1]].reshape(L + 1, M + 1, 8)[1:, 1:, :], dp[1:, 1:, :])
        L_.bo_gmx_free(g8)
        assert ok, (e, L)


HALF_SCRIPT = r"""
import sys, os
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import bath_amd as ba, oracle_lib as ol
from test_fs_chain_gpu import mixed_windows, check_fs3
path = ol.GOLDEN + "/Caudal_act.bhmm"
ctx = ba.Context(0)
model = ol.Model(path); hmm = ba.HMM(path)
om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3))
rng = np.random.default_rng({seed})
wins = mixed_windows(rng, model, {n_model}, {n_random}, lo=15, hi={hi})
check_fs3(ctx, model, om3, wins, "half-wave")
print("half-wave ok", len(wins))
"""


@pytest.mark.parametrize("switches,n_model,n_random,hi", [
    ({"BATH_HIP_FS_HALFWAVE": "1", "BATH_HIP_FS_BWD_HALFWAVE": "1"}, 200, 700, 420),                          # half-wave kernels, batches by length
    ({"BATH_HIP_FS_HALFWAVE": "1", "BATH_HIP_FS_BWD_HALFWAVE": "1", "BATH_HIP_FS_BATCH": "3"}, 60, 140, 600),   # ... uniform batches of 3
    ({"BATH_HIP_FS_FULLWAVE": "1", "BATH_HIP_FS_BWD_HALFWAVE": "0"}, 500, 1750, 380),                          # full-wave kernels at W = 16 windows per block
])
def test_fs3_chain_kernel_variants_are_bit_identical(switches, n_model, n_random, hi):
    """fs3_fwd_chain_half_kernel / fs3_bwd_chain_half_kernel (32 windows per block, every chain lane a row) forced onto a few
    hundred windows, and the full-wave kernels forced onto enough windows for 16 per block; fresh processes (static switches)."""
    env = dict(os.environ, **switches)
    r = subprocess.run([sys.executable, "-c", HALF_SCRIPT.format(root=ROOT, seed=5 + len(switches), n_model=n_model, n_random=n_random, hi=hi)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "half-wave ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


MEM_SCRIPT = r"""
import sys, os
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import bath_amd as ba, oracle_lib as ol, common
from test_fs_chain_gpu import mixed_windows, check_fs3
path = common.write_synthetic_bhmm({path!r}, {M}, seed={M})
ctx = ba.Context(0)
model = ol.Model(path); hmm = ba.HMM(path)
om3 = ba.FSOProfile(ctx, ba.FSProfile(hmm, 3))
rng = np.random.default_rng({M})
wins = mixed_windows(rng, model, 14, 50, lo=15, hi=520)
wins += [rng.integers(0, 4, size=L).astype(np.uint8) for L in (15, 16, 17, 18, 19, 20)]
check_fs3(ctx, model, om3, wins, "history in memory")
print("mem kernel ok", len(wins))
"""


@pytest.mark.parametrize("M,extra", [(700, {}), (1000, {}), (1024, {}), (1024, {"BATH_HIP_FS_BWD_WPW": "1"}), (1024, {"BATH_HIP_FS_FWD_MEM_GRID": "3"})])
def test_fs3_forward_chain_with_the_history_in_memory_is_bit_identical(tmp_path, M, extra):
    """fs3_fwd_chain_mem_kernel (long models, eight windows per block, the rows' history in global memory: what configs[4] takes from
    250 Mb on) forced onto ~80 windows of 15..520 nt with BATH_HIP_FS_FWD_MEM=1: 12 and 16 nodes per lane (1000: node slots past the
    model's end), batches whose windows end at different rows, windows too short for a codon.  Scores and special-state rows against
    the oracle, bit for bit.  The script checks the Backward parser on the same windows: the long models' kernel with two waves per
    window (the default up to four windows per block), and with BATH_HIP_FS_BWD_WPW=1 the one-wave instantiation that blocks of eight
    windows take.  BATH_HIP_FS_FWD_MEM_GRID=3: three blocks for the ten batches, so that a block's history records are reused by the
    windows of its next batches (configs[4] at 1 Gb: 330 batches on 256 blocks)."""
    env = dict(os.environ, BATH_HIP_FS_FWD_MEM="1", **extra)
    r = subprocess.run([sys.executable, "-c", MEM_SCRIPT.format(root=ROOT, path=str(tmp_path / ("s%d.bhmm" % M)), M=M)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mem kernel ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
