"""The C surface a C host drives several GPUs with (include/bath_hip.h: bath_hits_serialize / _deserialize,
bath_tophits_add_serialized, bath_dist_items / _deal / _shard_range), exercised FROM C: tests/c/multi_gpu_surface.c is compiled
with gcc against the header and linked with libbathhip.so, round-trips a hit list byte for byte, merges a remote rank's stream
into a hit list and prints its division of a 12-query job, which must be the one bath_amd.dist computes (dist.py calls the same
functions) and the one the rule written out in Python gives.  No GPU call anywhere."""
import os
import subprocess

import bath_amd as ba
from bath_amd import dist as bd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_round_trips_hits_and_divides_the_job(tmp_path):
    exe = str(tmp_path / "multi_gpu_surface")
    libdir = os.path.join(ROOT, "bath_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "multi_gpu_surface.c"),
                           "-L", libdir, "-lbathhip", "-Wl,-rpath," + libdir, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "multi-GPU C surface ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    lines = r.stdout.split("\n")
    got = [tuple(int(x) for x in l.split()[1:]) for l in lines if l.startswith("item ")]
    M = [78, 185, 56, 209, 218, 109, 153, 247, 220, 226, 90, 459]
    nwin = [382] * 12
    cost = [382.0 * (m + 150) for m in M]
    items = bd.query_items_weighted(nwin, cost, 8, items_per_rank=3)
    owner = bd.deal([(hi - lo) * (M[q] + 150.0) for q, lo, hi in items], 8)
    assert got == [(q, lo, hi, o) for (q, lo, hi), o in zip(items, owner)]
    # ... and the rule itself, written out: round(share x T) groups per query, LPT onto the least loaded rank
    T, total = max(12, 3 * 8), sum(cost)
    want = []
    for q in range(12):
        g = max(1, min(int(cost[q] / total * T + 0.5), nwin[q]))
        per, rem = divmod(nwin[q], g)
        for k in range(g):
            lo = k * per + min(k, rem)
            want.append((q, lo, lo + per + (1 if k < rem else 0)))
    assert [x[:3] for x in got] == want and len(want) > 12
    load = [0.0] * 8
    for x in sorted(range(len(want)), key=lambda x: (-(want[x][2] - want[x][1]) * (M[want[x][0]] + 150.0), x)):
        r_ = min(range(8), key=lambda k: (load[k], k))
        assert got[x][3] == r_
        load[r_] += (want[x][2] - want[x][1]) * (M[want[x][0]] + 150.0)
    assert [tuple(int(x) for x in l.split()[1:]) for l in lines if l.startswith("shard ")] == [(0, 0, 4), (1, 4, 7), (2, 7, 10)]


def test_python_hit_arrays_travel_as_the_c_stream():
    d = ba.FsDomain(); d.window = 5; d.reported = 1; d.iali = 10; d.jali = 100; d.lnP = -50.0; d.bitscore = 60.0; d.cigar = "30M"
    e = ba.FsDomain(); e.window = 6; e.reported = 1; e.iali = 400; e.jali = 100; e.strand = 1; e.lnP = -20.5; e.bitscore = 31.0; e.cigar = "12M3I9M"
    h = ba.HitArray.from_domains([d, e])
    b = h.to_bytes()
    assert b[:4] == b"BHIT" and int.from_bytes(b[8:16], "big") == 2               # network byte order, as p7_hit_Serialize writes
    h2, p = ba.HitArray.from_bytes(b + b"tail", 0)
    assert p == len(b) and h2.to_bytes() == b and h2.pool == b"30M\0" + b"12M3I9M\0"
    both = ba.HitArray.concat([h2, h])
    assert both.to_bytes() == ba.HitArray.from_bytes(both.to_bytes())[0].to_bytes() and len(both) == 4


def test_concat_keeps_absent_cigars_absent():
    """A hit shipped without a CIGAR comes back with cigar_off == -1 (bath_hits_deserialize); behind another part's pool it must stay
    -1, not become an offset into that pool (an absent CIGAR would silently turn into the previous part's terminating NUL)."""
    d = ba.FsDomain(); d.window = 1; d.reported = 1; d.iali = 10; d.jali = 100; d.cigar = "30M"
    e = ba.FsDomain(); e.window = 2; e.reported = 1; e.iali = 50; e.jali = 90; e.cigar = "7M1F5M"
    with_cig = ba.HitArray.from_domains([d, e])
    bare = ba.HitArray(with_cig.rec.copy(), b"")                                     # the same records, no pool: serialized without CIGARs
    bare2, _ = ba.HitArray.from_bytes(bare.to_bytes())
    assert list(bare2.rec["cigar_off"]) == [-1, -1]
    both = ba.HitArray.concat([with_cig, bare2, with_cig])
    assert list(both.rec["cigar_off"]) == [0, 4, -1, -1, 11, 15] and both.pool == with_cig.pool * 2
    again, _ = ba.HitArray.from_bytes(both.to_bytes())                                # ... and the mixed array travels
    assert list(again.rec["cigar_off"]) == [0, 4, -1, -1, 11, 15] and again.pool == both.pool


def test_remote_hits_outside_the_search_are_refused():
    """bath_tophits_add_serialized checks every shifted window against the number of sequences before anything indexes the name and
    length arrays (a damaged stream or a wrong shift must be an error code, not an out-of-bounds read)."""
    import ctypes as C
    d = ba.FsDomain(); d.window = 3; d.reported = 1; d.iali = 10; d.jali = 100; d.lnP = -40.0; d.bitscore = 50.0; d.cigar = "30M"
    b = ba.HitArray.from_domains([d]).to_bytes()
    names = (C.c_char_p * 4)(b"a", b"b", b"c", b"d")
    lens = (C.c_int64 * 4)(1000, 1000, 1000, 1000)
    L = ba.lib()
    th = L.bath_tophits_create()
    EFORMAT = L.bath_hits_deserialize(b"XXXX" + b[4:], len(b), C.byref(C.c_void_p()))
    assert EFORMAT != ba.OK
    assert L.bath_tophits_add_serialized(th, b, len(b), 1, 4, 0, names, None, None, lens) == EFORMAT      # 3 + 1 = 4: past the end
    assert L.bath_tophits_add_serialized(th, b, len(b), -4, 4, 0, names, None, None, lens) == EFORMAT     # negative
    assert L.bath_tophits_count(th) == 0
    assert L.bath_tophits_add_serialized(th, b, len(b), 0, 4, 0, names, None, None, lens) == ba.OK and L.bath_tophits_count(th) == 1
    L.bath_tophits_destroy(th)
    lo, hi = bd.shard_range(10, 0, 0)                                                # world 0: an empty share, no SIGFPE
    assert (lo, hi) == (0, 0)


def test_damaged_hit_streams_never_crash_the_reader():
    """A stream that crossed a process boundary is untrusted input: 3000 random corruptions of a valid stream (bytes flipped, the
    stream cut, counts and sizes overwritten, garbage appended) through bath_hits_stream_size / bath_hits_deserialize /
    bath_tophits_add_serialized.  Every call must return -- OK with a self-consistent result, or an error code -- and nothing may be
    added to the hit list by a call that reports an error.  (tools/san_cpu.sh runs this under ASan + UBSan.)"""
    import ctypes as C
    import numpy as np
    rng = np.random.default_rng(99)
    doms = []
    for w in range(6):
        d = ba.FsDomain(); d.window = w % 4; d.reported = 1; d.iali = 10 + w; d.jali = 100 + w; d.lnP = -30.0 - w; d.bitscore = 40.0 + w
        d.cigar = "%dM%dI%dM" % (10 + w, 1 + w % 3, 20 + w)
        doms.append(d)
    good = ba.HitArray.from_domains(doms).to_bytes()
    L = ba.lib()
    names = (C.c_char_p * 4)(b"a", b"b", b"c", b"d")
    lens = (C.c_int64 * 4)(1000, 1000, 1000, 1000)
    n_ok = n_err = 0
    for trial in range(3000):
        b = bytearray(good)
        kind = trial % 6
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 1:
            b = b[:int(rng.integers(0, len(b)))]
        elif kind == 2:
            b[8:16] = int(rng.integers(0, 2 ** 40)).to_bytes(8, "big")                      # the hit count
        elif kind == 3:
            p = 16
            b[p:p + 4] = int(rng.integers(0, 2 ** 31)).to_bytes(4, "big")                   # the first record's size
        elif kind == 4:
            b += bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8))
        else:
            p = int(rng.integers(16, len(b) - 4)); b[p:p + 4] = b"\\xff\\xff\\xff\\xff"
        raw = bytes(b)
        buf = (C.c_uint8 * max(len(raw), 1)).from_buffer_copy(raw if raw else b"\\0")       # an exact-size heap copy: an overrun is ASan's to catch
        size = L.bath_hits_stream_size(C.addressof(buf), len(raw))
        assert size == -1 or 16 <= size <= len(raw)
        h = C.c_void_p()
        st = L.bath_hits_deserialize(C.addressof(buf), len(raw), C.byref(h))
        if st == ba.OK:
            assert h.value and 0 <= L.bath_hits_count(h) <= 6 + 40
            L.bath_hits_destroy(h)
        else:
            assert not h.value
        th = L.bath_tophits_create()
        st2 = L.bath_tophits_add_serialized(th, C.addressof(buf), len(raw), 0, 4, 0, names, None, None, lens)
        if st2 == ba.OK:
            n_ok += 1
            assert L.bath_tophits_count(th) <= 6 + 40
        else:
            n_err += 1
            assert L.bath_tophits_count(th) == 0
        L.bath_tophits_destroy(th)
    assert n_err > 1500 and n_ok > 0                                     # most corruptions are caught; flips inside float fields are still valid streams
