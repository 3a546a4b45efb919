"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's hot path (oracle/bath_oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(HERE, "golden")

NEVPARAM = 8
K = 20
KP = 29
NTRANS = 8


def build():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(d, f) for d in (ORACLE_DIR, os.path.join(ORACLE_DIR, "sse")) for f in os.listdir(d) if f.endswith((".c", ".h"))]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return so


class Hmm(C.Structure):
    _fields_ = [("M", C.c_int), ("max_length", C.c_int), ("ct", C.c_int), ("fsprob", C.c_float),
                ("t", C.POINTER(C.c_float)), ("mat", C.POINTER(C.c_float)), ("ins", C.POINTER(C.c_float)),
                ("compo", C.c_float * K), ("evparam", C.c_float * NEVPARAM),
                ("name", C.c_char * 128), ("acc", C.c_char * 64), ("consensus", C.c_char_p)]


class Profile(C.Structure):
    _fields_ = [("M", C.c_int), ("L", C.c_int), ("max_length", C.c_int), ("nj", C.c_float),
                ("tsc", C.POINTER(C.c_float)), ("rsc", C.POINTER(C.c_float)),
                ("xsc", (C.c_float * 2) * 4), ("evparam", C.c_float * NEVPARAM), ("compo", C.c_float * K)]


class FsProfile(C.Structure):
    _fields_ = [("M", C.c_int), ("L", C.c_int), ("max_length", C.c_int), ("codon_lengths", C.c_int),
                ("maxcodons", C.c_int), ("nj", C.c_float), ("fsprob", C.c_float),
                ("tsc", C.POINTER(C.c_float)), ("rsc", C.POINTER(C.c_float)),
                ("codons", C.POINTER(C.c_uint8)), ("indel_pos", C.POINTER(C.c_uint8)),
                ("xsc", (C.c_float * 2) * 4), ("evparam", C.c_float * NEVPARAM), ("compo", C.c_float * K)]


class OProfile(C.Structure):
    _fields_ = [("M", C.c_int), ("L", C.c_int), ("max_length", C.c_int), ("nj", C.c_float),
                ("rb", C.POINTER(C.c_uint8)),
                ("tbm_b", C.c_uint8), ("tec_b", C.c_uint8), ("tjb_b", C.c_uint8), ("base_b", C.c_uint8),
                ("bias_b", C.c_uint8), ("scale_b", C.c_float),
                ("rw", C.POINTER(C.c_int16)), ("tw", C.POINTER(C.c_int16)),
                ("xw", (C.c_int16 * 2) * 4), ("scale_w", C.c_float), ("base_w", C.c_int16), ("ddbound_w", C.c_int16),
                ("rf", C.POINTER(C.c_float)), ("tf", C.POINTER(C.c_float)), ("xf", (C.c_float * 2) * 4),
                ("evparam", C.c_float * NEVPARAM), ("compo", C.c_float * K),
                ("msc", C.POINTER(C.c_float)), ("tsc", C.POINTER(C.c_float))]


def aliscore_drops():
    """Test hook of the oracle: envelopes dropped so far by the aliscore < 0 rule (p7_domaindef.c:1072,1286), both branches."""
    return C.c_int.in_dll(lib(), "bo_aliscore_drops").value


class Bg(C.Structure):
    _fields_ = [("f", C.c_float * K), ("p1", C.c_float), ("t", (C.c_float * 3) * 2),
                ("e", (C.c_float * K) * 2), ("eo", (C.c_float * 2) * KP), ("pi", C.c_float * 3)]


class ScoreData(C.Structure):
    _fields_ = [("M", C.c_int), ("ssv_scores", C.POINTER(C.c_uint8)),
                ("prefix_lengths", C.POINTER(C.c_float)), ("suffix_lengths", C.POINTER(C.c_float))]


class Window(C.Structure):
    _fields_ = [("id", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("length", C.c_int32), ("score", C.c_float)]


class WindowList(C.Structure):
    _fields_ = [("w", C.POINTER(Window)), ("count", C.c_int), ("size", C.c_int)]


class Orf(C.Structure):
    _fields_ = [("start", C.c_int32), ("end", C.c_int32), ("n", C.c_int32), ("frame", C.c_int32), ("off", C.c_int64)]


class OrfBlock(C.Structure):
    _fields_ = [("orf", C.POINTER(Orf)), ("count", C.c_int), ("size", C.c_int),
                ("aa", C.POINTER(C.c_uint8)), ("aa_n", C.c_int64), ("aa_size", C.c_int64)]


class Gmx(C.Structure):
    _fields_ = [("M", C.c_int), ("L", C.c_int), ("nrows", C.c_int), ("nscells", C.c_int),
                ("dp", C.POINTER(C.c_float)), ("xmx", C.POINTER(C.c_float))]


class Pipeline(C.Structure):
    _fields_ = [("F1", C.c_double), ("F2", C.c_double), ("F3", C.c_double), ("F4", C.c_double),
                ("do_biasfilter", C.c_int), ("fs_pipe", C.c_int), ("minlen", C.c_int),
                ("nres", C.c_int64), ("n_orfs", C.c_int64), ("n_past_msv", C.c_int64), ("n_past_bias", C.c_int64),
                ("n_past_vit", C.c_int64), ("n_past_fwd", C.c_int64),
                ("pos_past_msv", C.c_int64), ("pos_past_bias", C.c_int64), ("pos_past_vit", C.c_int64),
                ("pos_past_fwd", C.c_int64),
                ("cells_msv", C.c_int64), ("cells_vit", C.c_int64), ("cells_fwd", C.c_int64), ("E", C.c_double), ("context", C.c_int32),
                # option state (bath_oracle.h): --nonull2, --fsonly, --strand, -m/-M (+ the codon table id), --incT, -T, --seed
                ("do_null2", C.c_int32), ("std_pipe", C.c_int32), ("strands", C.c_int32), ("initiator", C.c_int32), ("ct", C.c_int32),
                ("inc_by_E", C.c_int32), ("T", C.c_double), ("seed", C.c_uint32)]


def apply_opts(pli, ct, opts):
    """opts: dict of bo_pipeline option fields (do_null2, std_pipe, strands, initiator, inc_by_E, T, seed, F1..F4, do_biasfilter, minlen)."""
    pli.ct = ct
    for k, v in (opts or {}).items():
        assert hasattr(pli, k), k
        setattr(pli, k, v)


class FsDomain(C.Structure):
    """bo_fsdomain: a domain of the frameshift branch and its hit scores (oracle/fs_domaindef.c)."""
    _fields_ = [("ienv", C.c_int32), ("jenv", C.c_int32), ("iali", C.c_int32), ("jali", C.c_int32), ("ihmm", C.c_int32), ("jhmm", C.c_int32),
                ("envsc", C.c_float), ("oasc", C.c_float), ("domcorrection", C.c_float),
                ("dombias", C.c_float), ("bitscore", C.c_float), ("pre_score", C.c_float),
                ("lnP", C.c_double), ("reported", C.c_int32), ("n_shifted_codons", C.c_int32), ("trace_idx", C.c_int32)]


class DomTrace(C.Structure):
    """bo_domtrace: dom->tr, first to last match state (bath_oracle.h)."""
    _fields_ = [("N", C.c_int32), ("win_start", C.c_int32), ("orf_start", C.c_int32), ("frameshift", C.c_int32),
                ("st", C.POINTER(C.c_int8)), ("k", C.POINTER(C.c_int32)), ("i", C.POINTER(C.c_int32)), ("c", C.POINTER(C.c_int8)),
                ("pp", C.POINTER(C.c_float))]


def trace_arrays(idx):
    """(DomTrace fields, st, k, i, c, pp as numpy copies) of the oracle's trace <idx>."""
    t = lib().bo_traces_get(idx).contents
    n = t.N
    arr = lambda p, dt: np.ctypeslib.as_array(p, shape=(n,)).astype(dt).copy()
    return t, arr(t.st, np.int8), arr(t.k, np.int32), arr(t.i, np.int32), arr(t.c, np.int8), arr(t.pp, np.float32)


class OrfResult(C.Structure):
    _fields_ = [("strand", C.c_int32), ("frame", C.c_int32), ("start", C.c_int32), ("end", C.c_int32), ("n", C.c_int32),
                ("stage", C.c_int32), ("msv_status", C.c_int32), ("vit_status", C.c_int32),
                ("usc", C.c_float), ("nullsc", C.c_float), ("filtersc", C.c_float), ("vfsc", C.c_float),
                ("fwdsc", C.c_float), ("P", C.c_double)]


class FsWindow(C.Structure):
    """bo_fswindow: one DNA window of p7_pli_Frameshift (oracle/fs_pipeline.c)."""
    _fields_ = [("strand", C.c_int32), ("n", C.c_int32), ("length", C.c_int32), ("k", C.c_int32),
                ("orf_cnt", C.c_int32), ("k_min", C.c_int32), ("k_max", C.c_int32),
                ("tot_orfsc", C.c_float), ("nullsc", C.c_float), ("filtersc", C.c_float), ("fwdsc", C.c_float),
                ("P_tot", C.c_double), ("P_min", C.c_double), ("P_fs", C.c_double), ("P_null", C.c_double),
                ("branch", C.c_int32), ("ndom", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    u8p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    L.bo_hmmfile_read.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.POINTER(Hmm))]
    L.bo_hmmfile_count.argtypes = [C.c_char_p]
    L.bo_hmm_free.argtypes = [C.POINTER(Hmm)]
    L.bo_bg_create.argtypes = [C.POINTER(Bg)]
    L.bo_bg_setlength.argtypes = [C.POINTER(Bg), C.c_int]
    L.bo_bg_setfilter.argtypes = [C.POINTER(Bg), C.c_int, f32p]
    L.bo_bg_nullone.argtypes = [C.POINTER(Bg), C.c_int]; L.bo_bg_nullone.restype = C.c_float
    L.bo_bg_fs_nullone.argtypes = [C.POINTER(Bg), C.c_int]; L.bo_bg_fs_nullone.restype = C.c_float
    L.bo_bg_filterscore.argtypes = [C.POINTER(Bg), u8p, C.c_int]; L.bo_bg_filterscore.restype = C.c_float
    L.bo_bg_fs_filterscore.argtypes = [C.POINTER(Bg), u8p, C.c_int, u8p]; L.bo_bg_fs_filterscore.restype = C.c_float
    L.bo_profile_config.argtypes = [C.POINTER(Hmm), C.POINTER(Bg), C.c_int]; L.bo_profile_config.restype = C.POINTER(Profile)
    L.bo_profile_reconfig_length.argtypes = [C.POINTER(Profile), C.c_int]
    L.bo_profile_free.argtypes = [C.POINTER(Profile)]
    L.bo_fs_profile_config.argtypes = [C.POINTER(Hmm), C.POINTER(Bg), u8p, C.c_int, C.c_int]
    L.bo_fs_profile_config.restype = C.POINTER(FsProfile)
    L.bo_fs_profile_reconfig_length.argtypes = [C.POINTER(FsProfile), C.c_int]
    L.bo_fs_profile_reconfig_unihit.argtypes = [C.POINTER(FsProfile), C.c_int]
    L.bo_fs_profile_reconfig_multihit.argtypes = [C.POINTER(FsProfile), C.c_int]
    L.bo_fs_profile_free.argtypes = [C.POINTER(FsProfile)]
    L.bo_oprofile_convert.argtypes = [C.POINTER(Profile)]; L.bo_oprofile_convert.restype = C.POINTER(OProfile)
    L.bo_oprofile_reconfig_length.argtypes = [C.POINTER(OProfile), C.c_int]
    L.bo_oprofile_reconfig_msv_length.argtypes = [C.POINTER(OProfile), C.c_int]
    L.bo_oprofile_free.argtypes = [C.POINTER(OProfile)]
    L.bo_scoredata_create.argtypes = [C.POINTER(OProfile)]; L.bo_scoredata_create.restype = C.POINTER(ScoreData)
    L.bo_scoredata_free.argtypes = [C.POINTER(ScoreData)]
    for fn in (L.bo_gumbel_surv, L.bo_gumbel_invsurv, L.bo_exp_surv):
        fn.argtypes = [C.c_double] * 3; fn.restype = C.c_double
    L.bo_flogsum.argtypes = [C.c_float, C.c_float]; L.bo_flogsum.restype = C.c_float
    L.bo_flogsum_set_exact.argtypes = [C.c_int]
    L.bo_flogsum_table.restype = f32p
    for fn in (L.bo_ssvfilter, L.bo_msvfilter, L.bo_msvfilter_noSSV, L.bo_vitfilter):
        fn.argtypes = [u8p, C.c_int, C.POINTER(OProfile), f32p]
    L.bo_ssvfilter_bath.argtypes = [u8p, C.c_int, C.POINTER(OProfile), C.POINTER(ScoreData), C.POINTER(Bg), C.c_double, C.POINTER(WindowList)]
    L.bo_vitfilter_bath.argtypes = [u8p, C.c_int, C.POINTER(OProfile), C.POINTER(ScoreData), C.c_float, C.c_double, C.POINTER(WindowList), f32p]
    L.bo_forward_parser.argtypes = [u8p, C.c_int, C.POINTER(OProfile), f32p, f32p]
    L.bo_backward_parser.argtypes = [u8p, C.c_int, C.POINTER(OProfile), f32p, f32p, f32p]
    L.bo_gviterbi.argtypes = [u8p, C.c_int, C.POINTER(Profile), f32p]
    L.bo_gforward.argtypes = [u8p, C.c_int, C.POINTER(Profile), f32p]
    L.bo_profile_same_as_mf.argtypes = [C.POINTER(OProfile), C.POINTER(Profile)]; L.bo_profile_same_as_mf.restype = C.POINTER(Profile)
    L.bo_profile_same_as_vf.argtypes = [C.POINTER(OProfile), C.POINTER(Profile)]; L.bo_profile_same_as_vf.restype = C.POINTER(Profile)
    # oracle/sse: the SSE2 striped restatement of impl_sse (the CPU baseline)
    L.bs_oprofile_create.argtypes = [C.POINTER(OProfile)]; L.bs_oprofile_create.restype = C.c_void_p
    L.bs_oprofile_free.argtypes = [C.c_void_p]
    for fn in (L.bs_ssvfilter, L.bs_msvfilter, L.bs_vitfilter, L.bs_forward_parser):
        fn.argtypes = [u8p, C.c_int, C.c_void_p, f32p]
    L.bs_vitfilter_bath.argtypes = [u8p, C.c_int, C.c_void_p, C.POINTER(ScoreData), C.c_float, C.c_double, C.POINTER(WindowList), f32p]
    L.bo_pipeline_use_sse.argtypes = [C.c_int]
    L.bo_windowlist_init.argtypes = [C.POINTER(WindowList)]
    L.bo_windowlist_free.argtypes = [C.POINTER(WindowList)]
    L.bo_orfblock_init.argtypes = [C.POINTER(OrfBlock)]
    L.bo_orfblock_reuse.argtypes = [C.POINTER(OrfBlock)]
    L.bo_orfblock_free.argtypes = [C.POINTER(OrfBlock)]
    L.bo_translate_orfs.argtypes = [u8p, C.c_int, u8p, C.c_int, C.POINTER(OrfBlock)]
    L.bo_translate_orfs_init.argtypes = [u8p, C.c_int, u8p, u8p, C.c_int, C.c_int, C.POINTER(OrfBlock)]
    L.bo_gencode_initiators.argtypes = [C.c_int, C.c_int, u8p]
    L.bo_set_seed.argtypes = [C.c_uint32]
    L.bo_traces_get.argtypes = [C.c_int]; L.bo_traces_get.restype = C.POINTER(DomTrace)
    L.bo_gencode_basic.argtypes = [C.c_int, u8p]
    L.bo_revcomp.argtypes = [u8p, C.c_int, u8p]
    L.bo_pipeline_init.argtypes = [C.POINTER(Pipeline), C.c_int]
    L.bo_pipeline_window_fs.argtypes = [C.POINTER(Pipeline), C.POINTER(OProfile), C.POINTER(FsProfile), C.POINTER(ScoreData), C.POINTER(Bg),
                                        u8p, u8p, C.c_int, C.POINTER(C.POINTER(OrfResult)), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.POINTER(FsWindow)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bo_pipeline_window_fsdom.argtypes = [C.POINTER(Pipeline), C.POINTER(OProfile), C.POINTER(FsProfile), C.POINTER(FsProfile), C.POINTER(ScoreData),
                                           C.POINTER(Bg), u8p, u8p, C.c_int, C.POINTER(C.POINTER(OrfResult)), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                           C.POINTER(C.POINTER(FsWindow)), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                           C.POINTER(C.POINTER(FsDomain)), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bo_pipeline_window_hits.argtypes = [C.POINTER(Pipeline), C.POINTER(OProfile), C.POINTER(ScoreData), C.POINTER(Bg), u8p, u8p, C.c_int,
                                          C.POINTER(C.POINTER(OrfResult)), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                          C.POINTER(C.POINTER(FsDomain)), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bo_pipeline_window.argtypes = [C.POINTER(Pipeline), C.POINTER(OProfile), C.POINTER(ScoreData), C.POINTER(Bg),
                                     u8p, u8p, C.c_int, C.POINTER(C.POINTER(OrfResult)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    if hasattr(L, "bo_gmx_create"):
        L.bo_gmx_create.argtypes = [C.c_int] * 4; L.bo_gmx_create.restype = C.POINTER(Gmx)
        L.bo_gmx_free.argtypes = [C.POINTER(Gmx)]
        L.bo_gforward_fs.argtypes = [u8p, C.c_int, C.POINTER(FsProfile), C.POINTER(Gmx), C.c_int, f32p]
        L.bo_gbackward_fs.argtypes = [u8p, C.c_int, C.POINTER(FsProfile), C.POINTER(Gmx), f32p]
        L.bo_gforward_parser_fs3.argtypes = [u8p, C.c_int, C.POINTER(FsProfile), C.POINTER(Gmx), f32p]
        L.bo_gbackward_parser_fs3.argtypes = [u8p, C.c_int, C.POINTER(FsProfile), C.POINTER(Gmx), f32p]
        L.bo_gdecoding_fs.argtypes = [C.POINTER(FsProfile), C.POINTER(Gmx), C.POINTER(Gmx)]
        L.bo_goptacc_fs.argtypes = [C.POINTER(FsProfile), C.POINTER(Gmx), C.POINTER(Gmx), f32p]
        L.bo_gnull2_fs.argtypes = [C.POINTER(FsProfile), C.POINTER(Gmx), f32p]
    _lib = L
    return L


def u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def dsq_from(codes):
    """1-based digital sequence with sentinels, as a numpy uint8 array of length n+2."""
    a = np.empty(len(codes) + 2, dtype=np.uint8)
    a[0] = a[-1] = 255
    a[1:-1] = codes
    return a


DNA_SYMS = "ACGT-RYMKSWHBVDN*~"
AMINO_SYMS = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~"


def digitize_dna(s):
    lut = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(DNA_SYMS):
        lut[ord(ch)] = i
        lut[ord(ch.lower())] = i
    lut[ord("U")] = lut[ord("u")] = 3
    lut[ord("X")] = lut[ord("x")] = 15
    codes = lut[np.frombuffer(s.encode(), dtype=np.uint8)]
    assert (codes != 255).all(), "bad DNA symbol"
    return codes


def digitize_amino(s):
    lut = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(AMINO_SYMS):
        lut[ord(ch)] = i
        lut[ord(ch.lower())] = i
    codes = lut[np.frombuffer(s.encode(), dtype=np.uint8)]
    assert (codes != 255).all(), "bad amino symbol"
    return codes


def read_fasta(path):
    out, name, buf = [], None, []
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(buf)))
                name, buf = line[1:].split()[0], []
            else:
                buf.append(line)
    if name is not None:
        out.append((name, "".join(buf)))
    return out


class Model:
    """Everything the oracle derives from one .bhmm record."""

    def __init__(self, path, index=0, L=100):
        L_ = lib()
        hp = C.POINTER(Hmm)()
        st = L_.bo_hmmfile_read(path.encode(), index, C.byref(hp))
        assert st == 0, "cannot read %s[%d]: %d" % (path, index, st)
        self.hmm = hp
        self.M = hp.contents.M
        self.bg = Bg()
        L_.bo_bg_create(C.byref(self.bg))
        self.gm = L_.bo_profile_config(hp, C.byref(self.bg), L)
        self.om = L_.bo_oprofile_convert(self.gm)
        self.sd = L_.bo_scoredata_create(self.om)
        self.basic = np.zeros(64, dtype=np.uint8)
        assert L_.bo_gencode_basic(hp.contents.ct, u8(self.basic)) == 0
        L_.bo_bg_setfilter(C.byref(self.bg), self.M, self.om.contents.compo)
        self._fs = {}

    def fs(self, codon_lengths, L_amino=100):
        key = codon_lengths
        if key not in self._fs:
            self._fs[key] = lib().bo_fs_profile_config(self.hmm, C.byref(self.bg), u8(self.basic), codon_lengths, L_amino)
        return self._fs[key]

    def run_pipeline(self, seqs, fs_pipe=False, opts=None):
        """seqs: list of digitized DNA code arrays. Returns (Pipeline counters, list of OrfResult copies)."""
        L_ = lib()
        pli = Pipeline()
        L_.bo_pipeline_init(C.byref(pli), 1 if fs_pipe else 0)
        apply_opts(pli, self.hmm.contents.ct, opts)
        res = C.POINTER(OrfResult)()
        nres, alloc = C.c_int(0), C.c_int(0)
        per_seq = []
        for codes in seqs:
            d = dsq_from(codes)
            before = nres.value
            L_.bo_pipeline_window(C.byref(pli), self.om, self.sd, C.byref(self.bg), u8(self.basic), u8(d), len(codes),
                                  C.byref(res), C.byref(nres), C.byref(alloc))
            per_seq.append((before, nres.value))
        out = [res[i] for i in range(nres.value)]
        return pli, out, per_seq

    def run_pipeline_hits(self, seqs, contexts=None, E=None, opts=None):
        """The plain pipeline through domain definition: (Pipeline counters, FsDomain records, per-sequence ranges, skipped).
        contexts[i]: ESL_SQ.C of window i (leading nucleotides shared with the previous window of the same target)."""
        L_ = lib()
        pli = Pipeline()
        L_.bo_pipeline_init(C.byref(pli), 0)
        apply_opts(pli, self.hmm.contents.ct, opts)
        if E is not None:
            pli.E = E                                          # the reporting threshold (-E, p7_pipeline.c:147)
        res = C.POINTER(OrfResult)(); nres, alloc = C.c_int(0), C.c_int(0)
        dm = C.POINTER(FsDomain)(); ndm, dmalloc, nskip = C.c_int(0), C.c_int(0), C.c_int(0)
        per_d = []
        for i, codes in enumerate(seqs):
            d = dsq_from(codes)
            pli.context = 0 if contexts is None else int(contexts[i])
            d0 = ndm.value
            L_.bo_pipeline_window_hits(C.byref(pli), self.om, self.sd, C.byref(self.bg), u8(self.basic), u8(d), len(codes),
                                       C.byref(res), C.byref(nres), C.byref(alloc), C.byref(dm), C.byref(ndm), C.byref(dmalloc), C.byref(nskip))
            per_d.append((d0, ndm.value))
        return pli, [dm[i] for i in range(ndm.value)], per_d, nskip.value

    def run_pipeline_fsdom(self, seqs, contexts=None, E=None, opts=None):
        """run_pipeline_fs plus domain definition and hit scores for the windows that take the frameshift branch.
        contexts[i]: ESL_SQ.C of window i (as in run_pipeline_hits).

        Returns (Pipeline counters, FsWindow records, per-sequence window ranges, FsDomain records, per-sequence domain
        ranges, number of multi-domain regions skipped)."""
        L_ = lib()
        pli = Pipeline()
        L_.bo_pipeline_init(C.byref(pli), 1)
        apply_opts(pli, self.hmm.contents.ct, opts)
        if E is not None:
            pli.E = E
        gm3, gm5 = self.fs(3), self.fs(5)
        res = C.POINTER(OrfResult)(); nres, alloc = C.c_int(0), C.c_int(0)
        fw = C.POINTER(FsWindow)(); nfw, fwalloc = C.c_int(0), C.c_int(0)
        dm = C.POINTER(FsDomain)(); ndm, dmalloc, nskip = C.c_int(0), C.c_int(0), C.c_int(0)
        per_w, per_d = [], []
        for i, codes in enumerate(seqs):
            d = dsq_from(codes)
            pli.context = 0 if contexts is None else int(contexts[i])
            w0, d0 = nfw.value, ndm.value
            L_.bo_pipeline_window_fsdom(C.byref(pli), self.om, gm3, gm5, self.sd, C.byref(self.bg), u8(self.basic), u8(d), len(codes),
                                        C.byref(res), C.byref(nres), C.byref(alloc), C.byref(fw), C.byref(nfw), C.byref(fwalloc),
                                        C.byref(dm), C.byref(ndm), C.byref(dmalloc), C.byref(nskip))
            per_w.append((w0, nfw.value)); per_d.append((d0, ndm.value))
        return pli, [fw[i] for i in range(nfw.value)], per_w, [dm[i] for i in range(ndm.value)], per_d, nskip.value

    def run_pipeline_fs(self, seqs, opts=None):
        """The cascade with fs_pipe set plus the frameshift stage (oracle/fs_pipeline.c) on every window.

        Returns (Pipeline counters, ORF records, per-sequence ORF ranges, FsWindow records, per-sequence window ranges)."""
        L_ = lib()
        pli = Pipeline()
        L_.bo_pipeline_init(C.byref(pli), 1)
        apply_opts(pli, self.hmm.contents.ct, opts)
        gm3 = self.fs(3)
        res = C.POINTER(OrfResult)()
        nres, alloc = C.c_int(0), C.c_int(0)
        fw = C.POINTER(FsWindow)()
        nfw, fwalloc = C.c_int(0), C.c_int(0)
        per_seq, per_seq_w = [], []
        for codes in seqs:
            d = dsq_from(codes)
            b0, w0 = nres.value, nfw.value
            L_.bo_pipeline_window_fs(C.byref(pli), self.om, gm3, self.sd, C.byref(self.bg), u8(self.basic), u8(d), len(codes),
                                     C.byref(res), C.byref(nres), C.byref(alloc), C.byref(fw), C.byref(nfw), C.byref(fwalloc))
            per_seq.append((b0, nres.value)); per_seq_w.append((w0, nfw.value))
        return pli, [res[i] for i in range(nres.value)], per_seq, [fw[i] for i in range(nfw.value)], per_seq_w
