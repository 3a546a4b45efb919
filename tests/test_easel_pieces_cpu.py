"""The pieces of easel that the multi-domain branch restates, held against INDEPENDENT implementations of the
published algorithms.  easel itself is absent from /root/reference (an un-vendored submodule), so no easel-produced vector
exists to compare with; what can be checked without it is that both restatements -- the product's (bath_ensemble.hip, through
the bath_selftest_* hooks of include/bath_hip.h) and the oracle's (oracle/stotrace.c) -- compute exactly what the published
definitions say:

  * esl_randomness_CreateFast(seed) + esl_random() (p7_pipeline.c:140 creates the pipeline's generator that way): Bob Jenkins'
    96-bit mix of (seed, 87654321, 12345678) from lookup2.c seeds the linear congruential generator x <- 69069 x + 1 (mod 2^32)
    of Marsaglia's "Super-Duper"; esl_random() = x / 2^32 in double.
  * esl_vec_FNorm (compensated float sum, then a division per element) followed by esl_rnd_FChoose (the first index whose running
    float sum exceeds the roll; a fresh roll if none does), as p7_StochasticTrace calls them (stotrace.c:165-300).

The implementations below are written from those definitions in plain Python / numpy float32, sharing no code with either
restatement."""
import ctypes as C

import numpy as np

import bath_amd as ba
import oracle_lib as ol

M32 = 0xFFFFFFFF


def jenkins_mix(a, b, c):
    """mix(a,b,c) of Bob Jenkins' lookup2.c (public domain, 1996): nine subtract-xor-shift rounds on 32-bit words."""
    for s1, s2, s3 in ((13, 8, 13), (12, 16, 5), (3, 10, 15)):
        a = (a - b - c) & M32; a ^= c >> s1
        b = (b - c - a) & M32; b ^= (a << s2) & M32
        c = (c - a - b) & M32; c ^= b >> s3
    return a, b, c


def fast_stream(seed, n):
    x = jenkins_mix(seed & M32, 87654321, 12345678)[2]
    if x == 0:
        x = 42
    out = np.empty(n, np.float64)
    for i in range(n):
        x = (x * 69069 + 1) & M32
        out[i] = x / 4294967296.0
    return out


def test_super_duper_recurrence_known_values():
    """x <- 69069 x + 1 (mod 2^32) from x = 0: 1, 69070, 475628535, 3277404108 -- the textbook head of Marsaglia's generator."""
    x, head = 0, []
    for _ in range(4):
        x = (x * 69069 + 1) & M32
        head.append(x)
    assert head == [1, 69070, 475628535, 3277404108]


def test_fast_generator_streams():
    L = ba.lib()
    for seed in (42, 1, 7, 0xFFFFFFFF, 123456789):
        want = fast_stream(seed, 3000)
        got = np.zeros(3000, np.float64)
        assert L.bath_selftest_rng_stream(seed, 3000, got.ctypes.data_as(C.POINTER(C.c_double))) == 0
        assert np.array_equal(got, want), seed                    # the product's generator (bath_ensemble.hip)
        O = ol.lib()
        O.bo_rng_next.restype = C.c_double
        r = (C.c_uint32 * 1)()
        O.bo_rng_init(r, C.c_uint32(seed))
        assert [O.bo_rng_next(r) for _ in range(200)] == list(want[:200]), seed        # the oracle's (oracle/stotrace.c)
    assert 0.0 <= fast_stream(42, 3000).min() and fast_stream(42, 3000).max() < 1.0


def fnorm_fchoose(stream, p):
    """esl_vec_FNorm then esl_rnd_FChoose in float32, consuming rolls from <stream> (an iterator of doubles)."""
    f32 = np.float32
    s, c = f32(0), f32(0)
    for x in p:
        y = f32(f32(x) - c); t = f32(s + y); c = f32(f32(t - s) - y); s = t
    q = [f32(f32(x) / s) if s != 0 else f32(1.0 / len(p)) for x in p]
    while True:
        roll = f32(next(stream))
        acc = f32(0)
        for i, x in enumerate(q):
            acc = f32(acc + x)
            if roll < acc:
                return i


def test_fnorm_and_fchoose():
    L = ba.lib()
    rng = np.random.default_rng(3)
    for n in (2, 4, 5):
        for trial in range(4):
            p = rng.random(n).astype(np.float32) ** 3 + np.float32(1e-6)
            if trial == 3:
                p[rng.integers(0, n)] = 0.0                       # an impossible choice must never be taken
            draws = 4000
            got = np.zeros(draws, np.int32)
            assert L.bath_selftest_fchoose(42, p.ctypes.data_as(C.POINTER(C.c_float)), n, draws, got.ctypes.data_as(C.POINTER(C.c_int32))) == 0
            stream = iter(fast_stream(42, draws * 4))
            want = [fnorm_fchoose(stream, p) for _ in range(draws)]
            assert list(got) == want, (n, trial)
            freq = np.bincount(got, minlength=n) / draws
            assert np.abs(freq - p / p.sum()).max() < 0.03            # and it samples the distribution
            assert not np.any(p[got] == 0.0)
