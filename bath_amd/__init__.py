"""bath_amd -- MI355X (gfx950) backend for BATH's bathsearch DP hot path.

Thin Python mirror of the C ABI in include/bath_hip.h (the product is libbathhip.so: hand-written HIP
kernels behind `extern "C"` entry points).  Names follow the reference's objects:

    HMM          P7_HMM            (p7_hmmfile.c:1342)
    Profile      P7_PROFILE        (modelconfig.c:48  p7_ProfileConfig)
    FSProfile    P7_FS_PROFILE     (modelconfig.c:220 p7_ProfileConfig_fs)
    OProfile     P7_OPROFILE       (impl_sse/p7_oprofile.c:1091 p7_oprofile_Convert), device resident
    FSOProfile   P7_FS_OPROFILE    (impl_sse/p7_fs_oprofile.c:221), device resident
    SeqBlock     ESL_SQ_BLOCK      digital sequences, device resident
    Pipeline     P7_PIPELINE       (p7_pipeline.c:94, :1584), the filter cascade

There is no CPU fallback: importing works anywhere (so the CPU test tier can check the library
loads and exports every symbol), but creating a Context without a usable GPU raises.
"""
import os as _os
# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read once when the runtime starts.  A context of this
# library owns up to a dozen streams (lanes, side and tail streams, the standard branch, the regions' Forward, the clusters' envelopes),
# and a host may run several contexts: with 4 queues, streams that are meant to run side by side end up one behind the other (two worker
# contexts on whole --fs passes: 53-58 ms per block with 4, 49-51 with 16; nine contexts on configs[3]: 18.0 -> 14.5 ms per database
# pass; one context alone: no difference).  Set before the first HIP call of the process; a value chosen by the caller is respected.
def _hip_runtime_initialised():
    """True only when this process has already made a HIP call that started the runtime (the case in which GPU_MAX_HW_QUEUES comes too
    late).  A mapped libamdhip64 says nothing -- `import torch` maps it long before any HIP call -- so the question goes to torch, the
    only way the runtime starts in a Python host before this module: torch.cuda.is_initialized() does not itself touch the GPU."""
    import sys as _sys
    t = _sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


HW_QUEUES_SET_BY_IMPORT = "GPU_MAX_HW_QUEUES" not in _os.environ
if HW_QUEUES_SET_BY_IMPORT:
    # (the variable is inherited by child processes; a host that wants another value sets it before importing this module)
    if _hip_runtime_initialised():
        import warnings as _warnings
        _warnings.warn("bath_amd: the HIP runtime is already initialised in this process, so GPU_MAX_HW_QUEUES=16 comes too late to take effect "
                       "(set it in the environment before the first HIP call: several worker contexts share 4 hardware queues otherwise; "
                       "INTEGRATION.md, 'Several contexts on one GPU')", RuntimeWarning, stacklevel=2)
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("BATH_HIP_LIBRARY") or os.path.join(_HERE, "libbathhip.so")   # the override is for A/B timing of two builds

OK, ERANGE, ENORESULT = 0, 16, 19
KP, K, NEVPARAM = 29, 20, 8
LOGSUM_TABLE, LOGSUM_EXACT, LOGSUM_TABLE_SERIAL, LOGSUM_CONTEXT = 0, 1, 2, 3

DNA_SYMS = "ACGT-RYMKSWHBVDN*~"
AMINO_SYMS = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~"


class BathError(RuntimeError):
    pass


def build(force=False):
    """Compile libbathhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcdir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(srcdir, f) for f in os.listdir(srcdir) if f.endswith((".hip", ".cpp", ".hpp"))]
    srcs.append(os.path.join(_ROOT, "include", "bath_hip.h"))
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-j4", "-C", srcdir])
    return LIB_PATH


class _Hmm(C.Structure):
    _fields_ = [("M", C.c_int32), ("max_length", C.c_int32), ("ct", C.c_int32), ("fsprob", C.c_float),
                ("t", C.POINTER(C.c_float)), ("mat", C.POINTER(C.c_float)), ("ins", C.POINTER(C.c_float)),
                ("compo", C.c_float * K), ("evparam", C.c_float * NEVPARAM), ("name", C.c_char * 128),
                ("acc", C.c_char * 64), ("consensus", C.c_char_p), ("rf", C.c_char_p), ("cs", C.c_char_p)]


class _Profile(C.Structure):
    _fields_ = [("M", C.c_int32), ("L", C.c_int32), ("max_length", C.c_int32), ("nj", C.c_float),
                ("tsc", C.POINTER(C.c_float)), ("rsc", C.POINTER(C.c_float)), ("xsc", (C.c_float * 2) * 4),
                ("evparam", C.c_float * NEVPARAM), ("compo", C.c_float * K)]


class _FsProfile(C.Structure):
    _fields_ = [("M", C.c_int32), ("L", C.c_int32), ("max_length", C.c_int32), ("codon_lengths", C.c_int32),
                ("maxcodons", C.c_int32), ("nj", C.c_float), ("fsprob", C.c_float),
                ("tsc", C.POINTER(C.c_float)), ("rsc", C.POINTER(C.c_float)),
                ("codons", C.POINTER(C.c_uint8)), ("indel_pos", C.POINTER(C.c_uint8)),
                ("xsc", (C.c_float * 2) * 4), ("evparam", C.c_float * NEVPARAM), ("compo", C.c_float * K)]


class OProfileScalars(C.Structure):
    _fields_ = [("tbm_b", C.c_uint8), ("tec_b", C.c_uint8), ("tjb_b", C.c_uint8), ("base_b", C.c_uint8), ("bias_b", C.c_uint8),
                ("scale_b", C.c_float), ("xw", (C.c_int16 * 2) * 4), ("scale_w", C.c_float),
                ("base_w", C.c_int16), ("ddbound_w", C.c_int16), ("xf", (C.c_float * 2) * 4)]


class PipelineParams(C.Structure):
    _fields_ = [("F1", C.c_double), ("F2", C.c_double), ("F3", C.c_double), ("F4", C.c_double),
                ("do_biasfilter", C.c_int32), ("fs_pipe", C.c_int32), ("min_orf_len", C.c_int32), ("ncbi_table", C.c_int32),
                ("nres_before", C.c_int64),
                # option state (include/bath_hip.h): --nonull2, --fsonly, --strand, -m / -M, --incT, --seed, -T
                ("do_null2", C.c_int32), ("std_pipe", C.c_int32), ("strands", C.c_int32), ("initiator", C.c_int32),
                ("inc_by_E", C.c_int32), ("seed", C.c_int32), ("T", C.c_double)]


STRAND_BOTH, STRAND_TOPONLY, STRAND_BOTTOMONLY = 0, 1, 2
INIT_ANY, INIT_TABLE, INIT_AUG = 0, 1, 2


class OrfResult(C.Structure):
    _fields_ = [("window", C.c_int64), ("strand", C.c_int32), ("frame", C.c_int32), ("start", C.c_int32), ("end", C.c_int32),
                ("n", C.c_int32), ("stage", C.c_int32), ("msv_status", C.c_int32), ("vit_status", C.c_int32),
                ("usc", C.c_float), ("nullsc", C.c_float), ("filtersc", C.c_float), ("vfsc", C.c_float), ("fwdsc", C.c_float),
                ("P", C.c_double)]


ORF_RESULT_DTYPE = np.dtype([("window", "<i8"), ("strand", "<i4"), ("frame", "<i4"), ("start", "<i4"), ("end", "<i4"),
                             ("n", "<i4"), ("stage", "<i4"), ("msv_status", "<i4"), ("vit_status", "<i4"),
                             ("usc", "<f4"), ("nullsc", "<f4"), ("filtersc", "<f4"), ("vfsc", "<f4"), ("fwdsc", "<f4"),
                             ("P", "<f8")], align=True)


class PipelineStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("nres", "n_orfs", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd",
                                          "pos_past_msv", "pos_past_bias", "pos_past_vit", "pos_past_fwd",
                                          "cells_msv", "cells_vit", "cells_fwd")]


class Orf(C.Structure):
    """bath_orf (include/bath_hip.h): one ORF of the six-frame translation."""
    _fields_ = [("window", C.c_int64), ("strand", C.c_int32), ("frame", C.c_int32), ("start", C.c_int32), ("end", C.c_int32),
                ("n", C.c_int32), ("aa_off", C.c_int64)]


class FsWindow(C.Structure):
    """bath_fs_window (include/bath_hip.h): one DNA window of the frameshift stage."""
    _fields_ = [("window", C.c_int64), ("strand", C.c_int32), ("n", C.c_int32), ("length", C.c_int32),
                ("orf_cnt", C.c_int32), ("k_min", C.c_int32), ("k_max", C.c_int32),
                ("tot_orfsc", C.c_float), ("nullsc", C.c_float), ("filtersc", C.c_float), ("fwdsc", C.c_float),
                ("P_tot", C.c_double), ("P_min", C.c_double), ("P_fs", C.c_double), ("P_null", C.c_double),
                ("branch", C.c_int32)]


class FsDomain(C.Structure):
    """bath_fs_domain (include/bath_hip.h): a domain of the frameshift branch with the scores of its hit."""
    _fields_ = [("window", C.c_int64), ("strand", C.c_int32), ("fs_window", C.c_int32),
                ("ienv", C.c_int32), ("jenv", C.c_int32), ("iali", C.c_int32), ("jali", C.c_int32), ("ihmm", C.c_int32), ("jhmm", C.c_int32),
                ("envsc", C.c_float), ("oasc", C.c_float), ("domcorrection", C.c_float), ("dombias", C.c_float),
                ("bitscore", C.c_float), ("pre_score", C.c_float), ("lnP", C.c_double), ("reported", C.c_int32), ("n_shifted_codons", C.c_int32),
                ("n_stops", C.c_int32), ("pid", C.c_float), ("ali_columns", C.c_int32), ("cigar_off", C.c_int64)]
    cigar = ""       # filled by Pipeline.run_hits / run_frameshift_domains


_DTYPES = {}


class DomainTrace(C.Structure):
    """bath_domain_trace: where a domain's trace lies in the arrays of bath_hip_domain_traces."""
    _fields_ = [("off", C.c_int64), ("N", C.c_int32), ("win_start", C.c_int32), ("orf_start", C.c_int32), ("frameshift", C.c_int32)]


class AliDisplayOpts(C.Structure):
    _fields_ = [("hmm_name", C.c_char_p), ("seq_name", C.c_char_p), ("consensus", C.c_char_p), ("rf", C.c_char_p), ("cs", C.c_char_p),
                ("M", C.c_int32), ("sqfrom", C.c_int64), ("sqto", C.c_int64), ("textw", C.c_int32), ("show_frameline", C.c_int32),
                ("initiator", C.c_int32)]


class DistItem(C.Structure):
    """bath_dist_item: windows [lo, hi) of query <query>."""
    _fields_ = [("query", C.c_int32), ("lo", C.c_int64), ("hi", C.c_int64)]


T_M, T_D, T_I = 1, 2, 3


def _np_dtype(T):
    """numpy's record dtype of a ctypes structure, converted once (the conversion walks the fields: 70 us for FsDomain)."""
    d = _DTYPES.get(T)
    if d is None:
        d = _DTYPES[T] = np.dtype(T)
    return d


FS_DOMAIN_DTYPE = _np_dtype(FsDomain)      # bath_fs_domain as a numpy record dtype


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_float), ("launches", C.c_int64), ("cells", C.c_double), ("bytes", C.c_double)]


class HmmWindow(C.Structure):
    _fields_ = [("target", C.c_int64), ("n", C.c_int32), ("k", C.c_int32), ("length", C.c_int32), ("score", C.c_float)]


class StdResult(C.Structure):
    _fields_ = [("fwdsc", C.c_float), ("bcksc", C.c_float), ("oasc", C.c_float), ("fwd_status", C.c_int32), ("bck_status", C.c_int32),
                ("ok", C.c_int32), ("null2", C.c_float * KP)]


class Fs5Result(C.Structure):
    _fields_ = [("fwdsc", C.c_float), ("bcksc", C.c_float), ("oasc", C.c_float), ("null2", C.c_float * KP)]


# name -> (restype, argtypes); every symbol include/bath_hip.h declares
_vp, _u8p, _f32p, _i32p, _i64p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
ABI = {
    "bath_hmmfile_count": (C.c_int, [C.c_char_p]),
    "bath_hmmfile_read": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.POINTER(_Hmm))]),
    "bath_hmm_destroy": (None, [C.POINTER(_Hmm)]),
    "bath_gencode_basic": (C.c_int, [C.c_int, _u8p]),
    "bath_profile_config": (C.c_int, [C.POINTER(_Hmm), C.c_int, C.POINTER(C.POINTER(_Profile))]),
    "bath_profile_destroy": (None, [C.POINTER(_Profile)]),
    "bath_fs_profile_config": (C.c_int, [C.POINTER(_Hmm), _u8p, C.c_int, C.c_int, C.POINTER(C.POINTER(_FsProfile))]),
    "bath_fs_profile_destroy": (None, [C.POINTER(_FsProfile)]),
    "bath_hip_init": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "bath_hip_finalize": (None, [_vp]),
    "bath_hip_last_error": (C.c_char_p, [_vp]),
    "bath_hip_synchronize": (C.c_int, [_vp]),
    "bath_hip_stream": (_vp, [_vp]),
    "bath_hip_set_fs_strict": (C.c_int, [_vp, C.c_int]),
    "bath_hip_set_fs_serial": (C.c_int, [_vp, C.c_int]),
    "bath_hip_trim": (C.c_int, [_vp]),
    "bath_hip_kernel_times": (C.c_int, [_vp, C.c_int, C.POINTER(KernelTime)]),
    "bath_hip_oprofile_convert": (C.c_int, [_vp, C.POINTER(_Profile), C.POINTER(_vp)]),
    "bath_hip_oprofile_destroy": (None, [_vp]),
    "bath_hip_oprofile_M": (C.c_int, [_vp]),
    "bath_hip_oprofile_scalars": (C.c_int, [_vp, C.c_int, C.POINTER(OProfileScalars)]),
    "bath_hip_oprofile_get_ssv_scores": (C.c_int, [_vp, _u8p]),
    "bath_hip_oprofile_get_vit": (C.c_int, [_vp, C.POINTER(C.c_int16), C.POINTER(C.c_int16)]),
    "bath_hip_oprofile_get_fwd": (C.c_int, [_vp, _f32p, _f32p]),
    "bath_hip_seqs_create": (C.c_int, [_vp, _u8p, _i64p, C.c_int64, C.POINTER(_vp)]),
    "bath_hip_seqs_destroy": (None, [_vp]),
    "bath_hip_seqs_count": (C.c_int64, [_vp]),
    "bath_hip_seqs_set_context": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "bath_hip_host_alloc": (C.c_void_p, [C.c_size_t]),
    "bath_hip_host_free": (None, [C.c_void_p]),
    "bath_hip_seqs_create_packed": (C.c_int, [_vp, _i64p, C.c_int64, C.POINTER(_vp)]),
    "bath_hip_seqs_upload_packed": (C.c_int, [_vp, C.c_void_p, _i64p, _i32p, _u8p, C.c_int64]),
    "bath_hip_seqs_upload_wait": (C.c_int, [_vp]),
    "bath_hip_ssvfilter": (C.c_int, [_vp, _vp, _vp, _f32p, _i32p]),
    "bath_hip_msvfilter": (C.c_int, [_vp, _vp, _vp, _f32p, _i32p]),
    "bath_hip_vitfilter": (C.c_int, [_vp, _vp, _vp, _f32p, _i32p]),
    "bath_hip_forward_parser": (C.c_int, [_vp, _vp, _vp, _f32p, _i32p]),
    "bath_hip_bias_filter": (C.c_int, [_vp, _vp, _vp, _f32p, _f32p]),
    "bath_hip_fwdback_parser": (C.c_int, [_vp, _vp, _vp, _i64p, _f32p, _f32p, _i32p, _i32p, _f32p, _f32p]),
    "bath_hip_translate_orfs": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(C.POINTER(Orf)), _i64p, C.POINTER(_u8p)]),
    "bath_hip_translate_orfs_opts": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(Orf)), _i64p, C.POINTER(_u8p)]),
    "bath_gencode_initiators": (C.c_int, [C.c_int, C.c_int, _u8p]),
    "bath_tophits_set_score_thresholds": (None, [_vp, C.c_int, C.c_double, C.c_int, C.c_double]),
    "bath_search_space_residues": (C.c_int64, [C.c_int, C.c_double, C.c_int, C.c_int64]),
    "bath_pipeline_params_default": (None, [C.POINTER(PipelineParams), C.c_int]),
    "bath_hip_pipeline_filters": (C.c_int, [_vp, _vp, _vp, C.POINTER(PipelineParams), C.POINTER(PipelineStats),
                                            C.POINTER(C.POINTER(OrfResult)), _i64p]),
    "bath_hip_pipeline_frameshift": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(PipelineParams), C.POINTER(PipelineStats),
                                              C.POINTER(C.POINTER(OrfResult)), _i64p, C.POINTER(C.POINTER(FsWindow)), _i64p]),
    "bath_hip_oprofile_set_consensus": (C.c_int, [_vp, C.c_char_p]),
    "bath_hip_domain_cigars": (C.c_void_p, [_vp]),
    "bath_hip_domain_traces": (C.c_int, [_vp, C.POINTER(C.POINTER(DomainTrace)), _i64p, C.POINTER(C.POINTER(C.c_int8)), C.POINTER(_i32p),
                                         C.POINTER(_i32p), C.POINTER(C.POINTER(C.c_int8)), C.POINTER(_f32p)]),
    "bath_alidisplay_print": (C.c_int64, [C.POINTER(DomainTrace), C.POINTER(C.c_int8), _i32p, _i32p, C.POINTER(C.c_int8), _f32p,
                                          _u8p, C.c_int32, C.POINTER(_FsProfile), C.POINTER(_Profile), _u8p, C.POINTER(AliDisplayOpts),
                                          C.c_char_p, C.c_int64]),
    "bath_selftest_rng_stream": (C.c_int, [C.c_uint32, C.c_int, C.POINTER(C.c_double)]),
    "bath_selftest_fchoose": (C.c_int, [C.c_uint32, _f32p, C.c_int, C.c_int, _i32p]),
    "bath_hits_serialize": (C.c_int64, [C.c_void_p, C.c_int64, C.c_char_p, C.POINTER(DomainTrace), C.POINTER(C.c_int8), _i32p, _i32p, C.POINTER(C.c_int8), _f32p,
                                        C.c_void_p, C.c_int64]),
    "bath_hits_deserialize": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(_vp)]),
    "bath_hits_stream_size": (C.c_int64, [C.c_void_p, C.c_int64]),
    "bath_hits_destroy": (None, [_vp]),
    "bath_hits_count": (C.c_int64, [_vp]),
    "bath_hits_domains": (C.POINTER(FsDomain), [_vp]),
    "bath_hits_cigars": (C.c_void_p, [_vp, _i64p]),
    "bath_hits_traces": (C.c_int, [_vp, C.POINTER(C.POINTER(DomainTrace)), C.POINTER(C.POINTER(C.c_int8)), C.POINTER(_i32p), C.POINTER(_i32p),
                                   C.POINTER(C.POINTER(C.c_int8)), C.POINTER(_f32p)]),
    "bath_tophits_add_serialized": (C.c_int, [_vp, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                              C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]),
    "bath_dist_shard_range": (None, [C.c_int64, C.c_int, C.c_int, _i64p, _i64p]),
    "bath_dist_items": (C.c_int64, [_i64p, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.POINTER(DistItem), C.c_int64]),
    "bath_dist_deal": (C.c_int, [C.POINTER(C.c_double), C.c_int64, C.c_int, _i32p]),
    "bath_selftest_cluster_segments": (C.c_int, [C.c_int, _i32p, _i32p, _i32p, _i32p, _i32p, C.c_int, C.c_int, _i32p, C.c_int, _i32p]),
    "bath_selftest_fs_ensemble": (C.c_int, [C.c_int, _f32p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _f32p, _f32p, _i32p, C.c_int, _i32p]),
    "bath_tophits_create": (_vp, []),
    "bath_tophits_destroy": (None, [_vp]),
    "bath_tophits_add": (C.c_int, [_vp, C.POINTER(FsDomain), C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                   C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]),
    "bath_tophits_finalize": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_double]),
    "bath_tophits_count": (C.c_int64, [_vp]),
    "bath_tophits_reported": (C.c_int64, [_vp]),
    "bath_tophits_get": (C.c_int, [_vp, C.c_int64, C.POINTER(FsDomain), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "bath_tophits_targets": (C.c_int64, [_vp, C.c_int, C.c_int, C.c_char_p, C.c_int64]),
    "bath_tophits_domain_annotation": (C.c_int64, [_vp, C.c_int64, C.c_int, C.c_int, C.c_char_p, C.c_int64]),
    "bath_tophits_pipeline_statistics": (C.c_int64, [_vp, C.POINTER(PipelineStats), C.POINTER(PipelineParams), C.c_int64, C.c_int64, C.c_int64, C.c_char_p, C.c_int64]),
    "bath_tophits_set_inclusion": (None, [_vp, C.c_double]),
    "bath_tophits_tabular_targets": (C.c_int64, [_vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int64]),
    "bath_hip_pipeline_hits": (C.c_int, [_vp, _vp, _vp, C.POINTER(PipelineParams), C.c_double, C.POINTER(PipelineStats),
                                         C.POINTER(C.POINTER(FsDomain)), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "bath_hip_pipeline_frameshift_domains": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(PipelineParams), C.c_double, C.POINTER(PipelineStats),
                                                      C.POINTER(C.POINTER(FsWindow)), _i64p, C.POINTER(C.POINTER(FsDomain)), _i64p, _i64p]),
    "bath_hip_pipeline_timings": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_char_p), _f32p, _i64p]),
    "bath_hip_fsprofile_convert": (C.c_int, [_vp, C.POINTER(_FsProfile), C.POINTER(_vp)]),
    "bath_hip_fsprofile_destroy": (None, [_vp]),
    "bath_hip_fs3_forward_parser": (C.c_int, [_vp, _vp, _vp, C.c_int, _f32p, _f32p, _i64p]),
    "bath_hip_fs3_backward_parser": (C.c_int, [_vp, _vp, _vp, C.c_int, _f32p, _f32p, _i64p]),
    "bath_hip_vitfilter_bath": (C.c_int, [_vp, _vp, _vp, _f32p, C.c_double, _f32p, _i32p, C.POINTER(C.POINTER(HmmWindow)), _i64p]),
    "bath_hip_ssvfilter_bath": (C.c_int, [_vp, _vp, _vp, C.c_double, C.POINTER(C.POINTER(HmmWindow)), _i64p]),
    "bath_hip_forward_full": (C.c_int, [_vp, _vp, _vp, _i32p, C.c_int, _f32p, _i32p, _f32p, _f32p]),
    "bath_hip_std_envelopes": (C.c_int, [_vp, _vp, _vp, C.POINTER(StdResult), _f32p, _f32p, _f32p, _f32p]),
    "bath_hip_fs5_envelopes_x": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.POINTER(Fs5Result), _f32p, _f32p, _f32p, _f32p]),
    "bath_hip_fs5_forward_full": (C.c_int, [_vp, _vp, _vp, C.c_int, _f32p, _f32p, _f32p]),
    "bath_hip_fs5_envelopes": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.POINTER(Fs5Result), _f32p, _i64p, _f32p, _i64p]),
}

_lib = None


def lib():
    """Load libbathhip.so (raises BathError with the build hint if it is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BathError("libbathhip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(or `make -C bath_amd/csrc`). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in ABI.items():
            fn = getattr(L, name)          # AttributeError => the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _u8(a):
    return a.ctypes.data_as(_u8p)


def _f32(a):
    return None if a is None else a.ctypes.data_as(_f32p)


def _i64(a):
    return None if a is None else a.ctypes.data_as(_i64p)


def digitize(seq, syms):
    lut = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(syms):
        lut[ord(ch)] = i
        lut[ord(ch.lower())] = i
    if syms is DNA_SYMS:
        lut[ord("U")] = lut[ord("u")] = 3
        lut[ord("X")] = lut[ord("x")] = 15
    codes = lut[np.frombuffer(seq.encode(), dtype=np.uint8)]
    if (codes == 255).any():
        raise BathError("symbol outside the alphabet")
    return codes


class HMM:
    """A profile HMM read from a BATH3/f file."""

    def __init__(self, path, index=0):
        p = C.POINTER(_Hmm)()
        st = lib().bath_hmmfile_read(os.fsencode(path), index, C.byref(p))
        if st != OK:
            raise BathError("cannot read model %d of %s (status %d)" % (index, path, st))
        self._p = p
        self.M = p.contents.M
        self.ct = p.contents.ct
        self.max_length = p.contents.max_length
        self.name = p.contents.name.decode()
        self.acc = p.contents.acc.decode()
        self.consensus = p.contents.consensus.decode() if p.contents.consensus else ""
        self.rf = p.contents.rf.decode() if p.contents.rf else None
        self.cs = p.contents.cs.decode() if p.contents.cs else None
        self.evparam = np.array(p.contents.evparam[:], dtype=np.float32)

    @staticmethod
    def count(path):
        return lib().bath_hmmfile_count(os.fsencode(path))

    def __del__(self):
        if getattr(self, "_p", None):
            lib().bath_hmm_destroy(self._p)
            self._p = None


def gencode_basic(ncbi_table):
    b = np.zeros(64, dtype=np.uint8)
    if lib().bath_gencode_basic(ncbi_table, _u8(b)) != OK:
        raise BathError("unknown NCBI translation table %d" % ncbi_table)
    return b


class Profile:
    """p7_ProfileConfig(hmm, bg, gm, L, p7_LOCAL)."""

    def __init__(self, hmm, L=100):
        p = C.POINTER(_Profile)()
        st = lib().bath_profile_config(hmm._p, L, C.byref(p))
        if st != OK:
            raise BathError("profile config failed (%d)" % st)
        self._p, self.M, self.hmm = p, hmm.M, hmm

    def arrays(self):
        g = self._p.contents
        tsc = np.ctypeslib.as_array(g.tsc, shape=(self.M, 8)).copy()
        rsc = np.ctypeslib.as_array(g.rsc, shape=(KP, self.M + 1, 2)).copy()
        xsc = np.array([[g.xsc[i][j] for j in range(2)] for i in range(4)], dtype=np.float32)
        return tsc, rsc, xsc

    def __del__(self):
        if getattr(self, "_p", None):
            lib().bath_profile_destroy(self._p)
            self._p = None


class FSProfile:
    """p7_ProfileConfig_fs(hmm, bg, gcode, gm_fs, L, p7_LOCAL) for 3 or 5 codon lengths."""

    def __init__(self, hmm, codon_lengths, L_amino=100, ncbi_table=None):
        self.basic = gencode_basic(hmm.ct if ncbi_table is None else ncbi_table)
        p = C.POINTER(_FsProfile)()
        st = lib().bath_fs_profile_config(hmm._p, _u8(self.basic), codon_lengths, L_amino, C.byref(p))
        if st != OK:
            raise BathError("fs profile config failed (%d)" % st)
        self._p, self.M, self.hmm, self.codon_lengths = p, hmm.M, hmm, codon_lengths
        self.maxcodons = p.contents.maxcodons

    def arrays(self):
        g = self._p.contents
        nrows = self.maxcodons + KP
        tsc = np.ctypeslib.as_array(g.tsc, shape=(self.M, 8)).copy()
        rsc = np.ctypeslib.as_array(g.rsc, shape=(nrows, self.M + 1)).copy()
        codons = np.ctypeslib.as_array(g.codons, shape=(self.M + 1, self.maxcodons)).copy()
        indel = np.ctypeslib.as_array(g.indel_pos, shape=(self.M + 1, self.maxcodons)).copy()
        return tsc, rsc, codons, indel

    def __del__(self):
        if getattr(self, "_p", None):
            lib().bath_fs_profile_destroy(self._p)
            self._p = None


class Context:
    """One GPU + one HIP stream (the reference's per-thread WORKER_INFO, bathsearch.c:34)."""

    def __init__(self, device=0):
        h = _vp()
        st = lib().bath_hip_init(device, C.byref(h))
        if st != OK:
            raise BathError("bath_hip_init(device=%d) failed with status %d: no usable gfx950 GPU; "
                            "this backend has no CPU fallback" % (device, st))
        self._h = h

    def _check(self, st, what):
        if st != OK:
            raise BathError("%s failed (%d): %s" % (what, st, lib().bath_hip_last_error(self._h).decode()))

    def synchronize(self):
        self._check(lib().bath_hip_synchronize(self._h), "synchronize")

    def trim(self):
        """Release the lanes, side contexts and side streams of this context (re-created on demand): before it sits idle beside
        worker contexts.  Arrays returned by earlier calls on this context are invalid afterwards."""
        self._check(lib().bath_hip_trim(self._h), "trim")

    def set_fs_strict(self, on=True):
        """True (the library's default): frameshift log-sums along the model in the reference's serial order, bit-identical to the
        generic reference.  False: the fast mode (wavefront scans, scores within O(1e-3) nats)."""
        self._check(lib().bath_hip_set_fs_strict(self._h, 1 if on else 0), "set_fs_strict")

    def set_fs_serial(self, on):
        """Measurement aid: the envelopes' Backward wavefront after the Forward one instead of beside it (bath_hip_set_fs_serial)."""
        self._check(lib().bath_hip_set_fs_serial(self._h, -1 if on is None else (1 if on else 0)), "set_fs_serial")

    @property
    def stream(self):
        return lib().bath_hip_stream(self._h)

    def close(self):
        if getattr(self, "_h", None):
            lib().bath_hip_finalize(self._h)
            self._h = None

    def __del__(self):
        self.close()


class OProfile:
    """p7_oprofile_Convert(gm, om): the device-resident limited-precision profile."""

    def __init__(self, ctx, gm):
        h = _vp()
        ctx._check(lib().bath_hip_oprofile_convert(ctx._h, gm._p, C.byref(h)), "oprofile_convert")
        self.ctx, self._h, self.M, self.gm = ctx, h, gm.M, gm
        hmm = getattr(gm, "hmm", None)
        if hmm is not None and hmm.consensus:                 # om->consensus, copied by p7_oprofile_Convert
            ctx._check(lib().bath_hip_oprofile_set_consensus(h, hmm.consensus.encode()), "oprofile_set_consensus")

    def scalars(self, L):
        s = OProfileScalars()
        self.ctx._check(lib().bath_hip_oprofile_scalars(self._h, L, C.byref(s)), "oprofile_scalars")
        return s

    def ssv_scores(self):
        a = np.zeros((self.M + 1, KP), dtype=np.uint8)
        lib().bath_hip_oprofile_get_ssv_scores(self._h, _u8(a))
        return a

    def vit_arrays(self):
        rw = np.zeros((KP, self.M + 1), dtype=np.int16)
        tw = np.zeros((self.M + 1, 8), dtype=np.int16)
        lib().bath_hip_oprofile_get_vit(self._h, rw.ctypes.data_as(C.POINTER(C.c_int16)), tw.ctypes.data_as(C.POINTER(C.c_int16)))
        return rw, tw

    def fwd_arrays(self):
        rf = np.zeros((KP, self.M + 1), dtype=np.float32)
        tf = np.zeros((self.M + 1, 8), dtype=np.float32)
        lib().bath_hip_oprofile_get_fwd(self._h, _f32(rf), _f32(tf))
        return rf, tf

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().bath_hip_oprofile_destroy(self._h)
        self._h = None


class FSOProfile:
    """p7_fs_oprofile_Convert(gm_fs, om_fs)."""

    def __init__(self, ctx, gm_fs):
        h = _vp()
        ctx._check(lib().bath_hip_fsprofile_convert(ctx._h, gm_fs._p, C.byref(h)), "fsprofile_convert")
        self.ctx, self._h, self.M, self.gm = ctx, h, gm_fs.M, gm_fs

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().bath_hip_fsprofile_destroy(self._h)
        self._h = None


class SeqBlock:
    """A block of digital sequences resident in HBM.  <seqs>: list of uint8 code arrays, or (flat, offsets)."""

    def __init__(self, ctx, seqs, offsets=None):
        if offsets is None:
            lens = np.array([len(s) for s in seqs], dtype=np.int64)
            offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
            np.cumsum(lens, out=offsets[1:])
            flat = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs]) if len(seqs) else np.zeros(0, np.uint8)
        else:
            flat = np.ascontiguousarray(seqs, dtype=np.uint8)
            offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if flat.size == 0:
            flat = np.zeros(1, np.uint8)
        self.n = len(offsets) - 1
        self.lengths = np.diff(offsets)
        h = _vp()
        ctx._check(lib().bath_hip_seqs_create(ctx._h, _u8(flat), _i64(offsets), self.n, C.byref(h)), "seqs_create")
        self.ctx, self._h = ctx, h

    def set_context(self, context):
        """ESL_SQ.C per window: leading nucleotides shared with the previous window of the same target (or None)."""
        if context is None:
            self.ctx._check(lib().bath_hip_seqs_set_context(self._h, None), "seqs_set_context")
            return
        c = np.ascontiguousarray(context, dtype=np.int32)
        assert c.shape == (self.n,)
        self.ctx._check(lib().bath_hip_seqs_set_context(self._h, c.ctypes.data_as(C.POINTER(C.c_int32))), "seqs_set_context")

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().bath_hip_seqs_destroy(self._h)
        self._h = None


def pack2(flat, offsets):
    """Host side of a streamed block: (packed bytes, exception sequence indices, positions, codes) for 1-byte DNA codes."""
    flat = np.ascontiguousarray(flat, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    lens = np.diff(offsets)
    n = len(lens)
    bad = np.flatnonzero(flat[offsets[0]:offsets[-1]] > 3) + offsets[0]
    seq = np.searchsorted(offsets, bad, side="right") - 1
    pos = (bad - offsets[seq]).astype(np.int32)
    codes = flat[bad].copy()
    if n and (lens == lens[0]).all() and lens[0] % 4 == 0:                     # equal windows, whole bytes: one reshape
        q = (flat[offsets[0]:offsets[-1]] & 3).reshape(-1, 4)
        packed = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)
    else:
        parts = []
        for i in range(n):
            s = flat[offsets[i]:offsets[i + 1]] & 3
            pad = (-len(s)) % 4
            q = np.concatenate([s, np.zeros(pad, np.uint8)]).reshape(-1, 4)
            parts.append((q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8))
        packed = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    return packed, seq.astype(np.int64), pos, codes


class PinnedBuffer:
    """Page-locked host memory (bath_hip_host_alloc) viewed as a numpy uint8 array: uploads from it are asynchronous."""

    def __init__(self, nbytes):
        self.ptr = lib().bath_hip_host_alloc(max(int(nbytes), 1))
        if not self.ptr:
            raise BathError("cannot allocate %d bytes of page-locked memory" % nbytes)
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(int(nbytes), 1)).from_address(self.ptr))

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().bath_hip_host_free(self.ptr)
            self.ptr = None


class StreamedBlock(SeqBlock):
    """A block that is refilled from the host in 2-bit form (bath_hip_seqs_create_packed / _upload_packed / _upload_wait)."""

    def __init__(self, ctx, offsets):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.n = len(offsets) - 1
        self.lengths = np.diff(offsets)
        h = _vp()
        ctx._check(lib().bath_hip_seqs_create_packed(ctx._h, _i64(offsets), self.n, C.byref(h)), "seqs_create_packed")
        self.ctx, self._h = ctx, h

    def upload(self, packed, exc_seq=None, exc_pos=None, exc_code=None):
        """Queue the transfer on the copy stream (returns at once for a PinnedBuffer / page-locked array)."""
        arr = packed.array if isinstance(packed, PinnedBuffer) else np.ascontiguousarray(packed, dtype=np.uint8)
        n_exc = 0 if exc_seq is None else len(exc_seq)
        es = np.ascontiguousarray(exc_seq if n_exc else np.zeros(1), dtype=np.int64)
        ep = np.ascontiguousarray(exc_pos if n_exc else np.zeros(1), dtype=np.int32)
        ec = np.ascontiguousarray(exc_code if n_exc else np.zeros(1), dtype=np.uint8)
        self._keep = (arr, es, ep, ec)
        self.ctx._check(lib().bath_hip_seqs_upload_packed(self._h, arr.ctypes.data, _i64(es), ep.ctypes.data_as(_i32p), _u8(ec), n_exc), "seqs_upload_packed")

    def wait(self):
        """Order the cascade after the upload and expand the block on the device."""
        self.ctx._check(lib().bath_hip_seqs_upload_wait(self._h), "seqs_upload_wait")


def translate_orfs(ctx, dna, ncbi_table=1, min_orf_len=20, strands=STRAND_BOTH, initiator=INIT_ANY):
    """Six-frame translation of a DNA SeqBlock (esl_gencode_Process*, bathsearch.c:384-392); strands / initiator: --strand, -m / -M.

    Returns a list of (window, strand, frame, start, end, residues[np.uint8]) sorted by window, strand, frame, start."""
    orfs = C.POINTER(Orf)()
    n = C.c_int64(0)
    aa = _u8p()
    ctx._check(lib().bath_hip_translate_orfs_opts(ctx._h, dna._h, ncbi_table, min_orf_len, strands, initiator, C.byref(orfs), C.byref(n), C.byref(aa)), "translate_orfs")
    out = []
    if n.value == 0:
        return out
    last = orfs[n.value - 1]
    pool = np.ctypeslib.as_array(aa, shape=(last.aa_off + last.n,))      # residues are stored in list order
    for i in range(n.value):
        o = orfs[i]
        out.append((o.window, o.strand, o.frame, o.start, o.end, pool[o.aa_off:o.aa_off + o.n].copy()))
    return out


def _score_call(fn, what, ctx, om, sq):
    sc = np.zeros(sq.n, dtype=np.float32)
    st = np.zeros(sq.n, dtype=np.int32)
    ctx._check(fn(ctx._h, om._h, sq._h, _f32(sc), st.ctypes.data_as(_i32p)), what)
    return sc, st


def SSVFilter(ctx, om, sq):
    """p7_SSVFilter over a block: (scores[n] nats, status[n])."""
    return _score_call(lib().bath_hip_ssvfilter, "ssvfilter", ctx, om, sq)


def MSVFilter(ctx, om, sq):
    """p7_MSVFilter over a block."""
    return _score_call(lib().bath_hip_msvfilter, "msvfilter", ctx, om, sq)


def ViterbiFilter(ctx, om, sq):
    """p7_ViterbiFilter over a block."""
    return _score_call(lib().bath_hip_vitfilter, "vitfilter", ctx, om, sq)


def ForwardParser(ctx, om, sq):
    """p7_ForwardParser over a block."""
    return _score_call(lib().bath_hip_forward_parser, "forward_parser", ctx, om, sq)


def FwdBackParser(ctx, om, sq):
    """p7_ForwardParser + p7_BackwardParser over a block: (fwd_sc[n], bck_sc[n], fwd_status[n], bck_status[n],
    [fwd xmx (L+1,6)], [bck xmx (L+1,6)]) with xmx columns {E,N,J,B,C,SCALE}."""
    n = sq.n
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum((sq.lengths + 1) * 6, out=offs[1:])
    fsc, bsc = np.zeros(n, np.float32), np.zeros(n, np.float32)
    fst, bst = np.zeros(n, np.int32), np.zeros(n, np.int32)
    fx, bx = np.zeros(max(int(offs[-1]), 1), np.float32), np.zeros(max(int(offs[-1]), 1), np.float32)
    ctx._check(lib().bath_hip_fwdback_parser(ctx._h, om._h, sq._h, _i64(offs), _f32(fsc), _f32(bsc), fst.ctypes.data_as(_i32p),
                                             bst.ctypes.data_as(_i32p), _f32(fx), _f32(bx)), "fwdback_parser")
    cut = lambda a: [a[offs[i]:offs[i + 1]].reshape(-1, 6) for i in range(n)]
    return fsc, bsc, fst, bst, cut(fx), cut(bx)


def BiasFilter(ctx, om, sq):
    """(p7_bg_NullOne, p7_bg_FilterScore) over a block."""
    nullsc = np.zeros(sq.n, dtype=np.float32)
    filtersc = np.zeros(sq.n, dtype=np.float32)
    ctx._check(lib().bath_hip_bias_filter(ctx._h, om._h, sq._h, _f32(nullsc), _f32(filtersc)), "bias_filter")
    return nullsc, filtersc


class Pipeline:
    """The filter cascade of p7_Pipeline_BATH over blocks of DNA windows."""

    def __init__(self, ctx, om, fs_pipe=False, ncbi_table=1, **overrides):
        self.ctx, self.om = ctx, om
        self.params = PipelineParams()
        lib().bath_pipeline_params_default(C.byref(self.params), 1 if fs_pipe else 0)
        self.params.ncbi_table = ncbi_table
        for k, v in overrides.items():
            setattr(self.params, k, v)

    def run(self, dna, want_results=True, copy=True):
        """copy=False: the records are a view of the library's page-locked result array, valid until the next pipeline call."""
        stats = PipelineStats()
        res = C.POINTER(OrfResult)()
        n = C.c_int64(0)
        self.ctx._check(lib().bath_hip_pipeline_filters(self.ctx._h, self.om._h, dna._h, C.byref(self.params), C.byref(stats),
                                                        C.byref(res) if want_results else None, C.byref(n)), "pipeline_filters")
        out = None
        if want_results:
            if n.value:
                buf = (OrfResult * n.value).from_address(C.addressof(res.contents))
                out = np.frombuffer(buf, dtype=ORF_RESULT_DTYPE)
                if copy:
                    out = out.copy()
            else:
                out = np.zeros(0, dtype=ORF_RESULT_DTYPE)
        return stats, out

    def run_frameshift(self, om_fs3, dna):
        """bathsearch --fs up to the branch decision: (stats, ORF records, list of FsWindow copies)."""
        stats = PipelineStats()
        res = C.POINTER(OrfResult)()
        n = C.c_int64(0)
        fw = C.POINTER(FsWindow)()
        nfw = C.c_int64(0)
        self.ctx._check(lib().bath_hip_pipeline_frameshift(self.ctx._h, self.om._h, om_fs3._h, dna._h, C.byref(self.params), C.byref(stats),
                                                           C.byref(res), C.byref(n), C.byref(fw), C.byref(nfw)), "pipeline_frameshift")
        if n.value:
            buf = (OrfResult * n.value).from_address(C.addressof(res.contents))
            out = np.frombuffer(buf, dtype=ORF_RESULT_DTYPE).copy()
        else:
            out = np.zeros(0, dtype=ORF_RESULT_DTYPE)
        wins = []
        for i in range(nfw.value):
            w = FsWindow()
            C.memmove(C.byref(w), C.byref(fw[i]), C.sizeof(FsWindow))
            wins.append(w)
        return stats, out, wins

    def run_hits(self, dna, E_report=10.0, arrays=False, nres_before=0):
        """bathsearch (no --fs) through domain definition and hit scores: (stats, [FsDomain], multi-domain regions skipped).
        arrays=True: (stats, HitArray, regions) -- the records as ONE numpy record array and the CIGAR strings as one bytes pool,
        copied out of the library with two memcpys instead of a Python object per hit (what TopHits.add_arrays and
        dist.gather_query_hits take).  nres_before: the residues the search counted before this block's first window (both strands;
        pli->nres on entry) -- with it a search cut into blocks reports the hits of the same search run as one block."""
        stats = PipelineStats()
        self.params.nres_before = int(nres_before)
        dm = C.POINTER(FsDomain)(); ndm = C.c_int64(0)
        nskip = C.c_int64(0)
        self.ctx._check(lib().bath_hip_pipeline_hits(self.ctx._h, self.om._h, dna._h, C.byref(self.params), E_report, C.byref(stats),
                                                     C.byref(dm), C.byref(ndm), C.byref(nskip)), "pipeline_hits")
        if arrays:
            return stats, self._hit_array(dm, ndm.value), nskip.value
        return stats, self._domains(dm, ndm.value), nskip.value

    def _hit_array(self, dm, n):
        if n == 0:
            return HitArray(np.zeros(0, dtype=FS_DOMAIN_DTYPE), b"")
        rec = np.frombuffer((FsDomain * n).from_address(C.addressof(dm.contents)), dtype=FS_DOMAIN_DTYPE).copy()
        base = lib().bath_hip_domain_cigars(self.ctx._h)
        last = int(rec["cigar_off"].max())
        end = last + len(C.string_at(base + last)) + 1 if base else 0
        return HitArray(rec, C.string_at(base, end) if base else b"")

    def _domains(self, dm, n):
        """Copies of the ctx-owned domain records, each with its --cigar string attached."""
        base = lib().bath_hip_domain_cigars(self.ctx._h)
        out = []
        for i in range(n):
            x = FsDomain()
            C.memmove(C.byref(x), C.byref(dm[i]), C.sizeof(FsDomain))
            x.cigar = C.string_at(base + x.cigar_off).decode() if base else ""
            out.append(x)
        return out

    def run_frameshift_domains(self, om_fs3, om_fs5, dna, E_report=10.0, arrays=False, nres_before=0):
        """bathsearch --fs through domain definition: (stats, [FsWindow], [FsDomain], multi-domain regions skipped).
        arrays=True: the windows and domains as numpy record arrays viewing the library's own memory (valid until the next
        pipeline call), without the per-record Python objects and CIGAR strings.  nres_before: as in run_hits."""
        stats = PipelineStats()
        self.params.nres_before = int(nres_before)
        fw = C.POINTER(FsWindow)(); nfw = C.c_int64(0)
        dm = C.POINTER(FsDomain)(); ndm = C.c_int64(0)
        nskip = C.c_int64(0)
        self.ctx._check(lib().bath_hip_pipeline_frameshift_domains(self.ctx._h, self.om._h, om_fs3._h, om_fs5._h, dna._h, C.byref(self.params), E_report,
                                                                   C.byref(stats), C.byref(fw), C.byref(nfw), C.byref(dm), C.byref(ndm), C.byref(nskip)),
                        "pipeline_frameshift_domains")
        if arrays:
            def view(ptr, n, T):
                if n == 0:
                    return np.zeros(0, dtype=_np_dtype(T))
                return np.frombuffer((T * n).from_address(C.addressof(ptr.contents)), dtype=_np_dtype(T))
            return stats, view(fw, nfw.value, FsWindow), view(dm, ndm.value, FsDomain), nskip.value

        def copies(ptr, n, T):
            out = []
            for i in range(n):
                x = T()
                C.memmove(C.byref(x), C.byref(ptr[i]), C.sizeof(T))
                out.append(x)
            return out
        return stats, copies(fw, nfw.value, FsWindow), self._domains(dm, ndm.value), nskip.value

    def traces(self):
        """P7_DOMAIN.tr of every domain of the last run_hits / run_frameshift_domains call (bath_hip_domain_traces): a list of
        (DomainTrace copy, st, k, i, c, pp) with numpy copies of the five arrays, entry d for domain d of that call."""
        tr = C.POINTER(DomainTrace)(); n = C.c_int64(0)
        st = C.POINTER(C.c_int8)(); k = _i32p(); i = _i32p(); c = C.POINTER(C.c_int8)(); pp = _f32p()
        self.ctx._check(lib().bath_hip_domain_traces(self.ctx._h, C.byref(tr), C.byref(n), C.byref(st), C.byref(k), C.byref(i), C.byref(c), C.byref(pp)),
                        "domain_traces")
        out = []
        for d in range(n.value):
            t = DomainTrace()
            C.memmove(C.byref(t), C.byref(tr[d]), C.sizeof(DomainTrace))
            sl = slice(t.off, t.off + t.N)
            grab = lambda p, dt: np.ctypeslib.as_array(p, shape=(t.off + t.N,))[sl].astype(dt).copy() if t.N else np.zeros(0, dt)
            out.append((t, grab(st, np.int8), grab(k, np.int32), grab(i, np.int32), grab(c, np.int8), grab(pp, np.float32)))
        return out

    def kernel_times(self):
        """Per-kernel device times of the stages after the cascade in the last run_frameshift_domains call:
        {name: (ms, launches, cells, bytes)}."""
        arr = (KernelTime * 32)()
        n = lib().bath_hip_kernel_times(self.ctx._h, 32, arr)
        return {arr[i].name.decode(): (float(arr[i].ms), int(arr[i].launches), float(arr[i].cells), float(arr[i].bytes)) for i in range(n)}

    def timings(self):
        names = (C.c_char_p * 32)()
        ms = np.zeros(32, dtype=np.float32)
        launches = np.zeros(32, dtype=np.int64)
        k = lib().bath_hip_pipeline_timings(self.ctx._h, 32, names, _f32(ms), _i64(launches))
        return [(names[i].decode(), float(ms[i]), int(launches[i])) for i in range(k)]


def alidisplay_print(trace, window_codes, hmm, sqfrom, sqto, seq_name, gm_fs5=None, gm=None, ncbi_table=1, textw=150, frameline=False,
                     initiator=INIT_ANY):
    """The alignment block of one hit (bath_alidisplay_print: p7_alidisplay_*_Create + p7_alidisplay_Print_BATH).
    trace: one entry of Pipeline.traces() (or the same six things from another source); window_codes: the digital nucleotides of
    windowsq (the strand read, from trace[0].win_start on); gm_fs5 / gm: FSProfile (5 codon lengths) / Profile objects."""
    t, st, k, i, c, pp = trace
    st = np.ascontiguousarray(st, np.int8); k = np.ascontiguousarray(k, np.int32); i = np.ascontiguousarray(i, np.int32)
    c = np.ascontiguousarray(c, np.int8); pp = np.ascontiguousarray(pp, np.float32)
    win = np.ascontiguousarray(window_codes, np.uint8)
    tr = DomainTrace(0, int(t.N), int(t.win_start), int(t.orf_start), int(t.frameshift))
    o = AliDisplayOpts(hmm.name.encode(), seq_name.encode(), hmm.consensus.encode(), hmm.rf.encode() if hmm.rf else None,
                       hmm.cs.encode() if hmm.cs else None, hmm.M, int(sqfrom), int(sqto), textw, 1 if frameline else 0, initiator)
    basic = gencode_basic(ncbi_table)
    p8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int8))
    args = [C.byref(tr), p8(st), k.ctypes.data_as(_i32p), i.ctypes.data_as(_i32p), p8(c), _f32(pp), _u8(win), len(win),
            gm_fs5._p if gm_fs5 is not None else None, gm._p if gm is not None else None, _u8(basic), C.byref(o)]
    n = lib().bath_alidisplay_print(*args, None, 0)
    if n < 0:
        raise BathError("alidisplay_print failed")
    buf = C.create_string_buffer(n + 1)
    lib().bath_alidisplay_print(*args, buf, n)
    return buf.raw[:n].decode()


class HitArray:
    """The hits of a pipeline call as one numpy record array (dtype of FsDomain) plus the pool their cigar_off fields point into."""

    def __init__(self, rec, pool):
        self.rec, self.pool = rec, pool

    def __len__(self):
        return len(self.rec)

    def to_bytes(self):
        """The library's hit stream (bath_hits_serialize: self-delimiting records in network byte order, after p7_hit_Serialize):
        what dist.gather_query_hits ships and what a C host would ship."""
        rec = np.ascontiguousarray(self.rec)
        pool = self.pool if self.pool else None
        args = (rec.ctypes.data, len(rec), pool, None, None, None, None, None, None)
        n = lib().bath_hits_serialize(*args, None, 0)
        if n < 0:
            raise BathError("hits_serialize failed")
        buf = C.create_string_buffer(n)
        if lib().bath_hits_serialize(*args, C.addressof(buf), n) != n:
            raise BathError("hits_serialize failed")
        return buf.raw

    @staticmethod
    def from_bytes(buf, p=0):
        """(HitArray, position behind the stream) from the stream that starts at buf[p] (bath_hits_deserialize)."""
        view = bytes(buf[p:]) if p else bytes(buf)
        size = lib().bath_hits_stream_size(view, len(view))
        if size < 0:
            raise BathError("not a hit stream")
        h = _vp()
        if lib().bath_hits_deserialize(view, size, C.byref(h)) != OK:
            raise BathError("hits_deserialize failed")
        try:
            n = lib().bath_hits_count(h)
            rec = np.frombuffer((FsDomain * n).from_address(C.addressof(lib().bath_hits_domains(h).contents)), dtype=FS_DOMAIN_DTYPE).copy() if n else np.zeros(0, dtype=FS_DOMAIN_DTYPE)
            k = C.c_int64(0)
            base = lib().bath_hits_cigars(h, C.byref(k))
            pool = C.string_at(base, k.value) if base and k.value else b""
        finally:
            lib().bath_hits_destroy(h)
        return HitArray(rec, pool), p + size

    @staticmethod
    def from_domains(domains):
        """From FsDomain objects carrying .cigar (the object path of run_hits / run_frameshift_domains)."""
        rec = np.zeros(len(domains), dtype=FS_DOMAIN_DTYPE)
        pool = bytearray()
        for i, d in enumerate(domains):
            rec[i] = np.frombuffer(bytes(d), dtype=FS_DOMAIN_DTYPE)[0]
            rec[i]["cigar_off"] = len(pool)
            pool += d.cigar.encode() + b"\0"
        return HitArray(rec, bytes(pool))

    @staticmethod
    def concat(parts):
        """One array from several: every part's cigar offsets moved behind the pools before it."""
        if len(parts) == 1:
            return parts[0]
        recs, pools, shift = [], [], 0
        for h in parts:
            r = h.rec.copy()
            r["cigar_off"][r["cigar_off"] >= 0] += shift            # -1 = the hit came without a CIGAR: stays -1
            recs.append(r); pools.append(h.pool); shift += len(h.pool)
        return HitArray(np.concatenate(recs) if recs else np.zeros(0, dtype=FS_DOMAIN_DTYPE), b"".join(pools))


class TopHits:
    """P7_TOPHITS for this path: collect the hits of pipeline calls, finish the search (E-values, duplicates, sorting,
    thresholds; bathsearch.c:868-921) and print --tblout (p7_tophits_TabularTargets)."""

    def __init__(self):
        self._h = lib().bath_tophits_create()

    def add(self, domains, names, lengths, seqidx0=0, accs=None, descs=None):
        n = len(domains)
        arr = (FsDomain * max(n, 1))()
        pool = bytearray()
        for i, d in enumerate(domains):
            C.memmove(C.byref(arr[i]), C.byref(d), C.sizeof(FsDomain))
            arr[i].cigar_off = len(pool)
            pool += d.cigar.encode() + b"\0"
        cig = C.create_string_buffer(bytes(pool) + b"\0")

        def strs(v):
            if v is None:
                return None
            a = (C.c_char_p * len(v))()
            for i, x in enumerate(v):
                a[i] = x.encode() if x else None
            return a
        lens = (C.c_int64 * len(lengths))(*[int(x) for x in lengths])
        st = lib().bath_tophits_add(self._h, arr, n, C.cast(cig, C.c_void_p), seqidx0, strs(names), strs(accs), strs(descs), lens)
        if st != OK:
            raise BathError("tophits_add failed (%d)" % st)

    def add_arrays(self, hits, names, lengths, seqidx0=0):
        """add() for a HitArray: the record array and the CIGAR pool go to the library as they are."""
        n = len(hits)
        if n == 0:
            return
        rec = np.ascontiguousarray(hits.rec)
        a = (C.c_char_p * len(names))(*[x.encode() for x in names])
        lens = (C.c_int64 * len(lengths))(*[int(x) for x in lengths])
        pool = C.create_string_buffer(hits.pool + b"\0")
        st = lib().bath_tophits_add(self._h, rec.ctypes.data_as(C.POINTER(FsDomain)), n, C.cast(pool, C.c_void_p), seqidx0, a, None, None, lens)
        if st != OK:
            raise BathError("tophits_add failed (%d)" % st)

    def reported(self):
        return int(lib().bath_tophits_reported(self._h))

    def set_score_thresholds(self, by_E=True, T=0.0, inc_by_E=True, incT=0.0):
        """-T / --incT: report / include by bit score instead of E-value (p7_pli_TargetReportable / Includable)."""
        lib().bath_tophits_set_score_thresholds(self._h, 1 if by_E else 0, T, 1 if inc_by_E else 0, incT)

    def finalize(self, nres, max_length, E=10.0):
        if lib().bath_tophits_finalize(self._h, nres, max_length, E) != OK:
            raise BathError("tophits_finalize failed")

    def hits(self):
        out = []
        for r in range(lib().bath_tophits_count(self._h)):
            d, idx, fl = FsDomain(), C.c_int64(0), C.c_int32(0)
            lib().bath_tophits_get(self._h, r, C.byref(d), C.byref(idx), C.byref(fl))
            out.append((d, idx.value, fl.value))
        return out

    def tblout(self, qname, qacc, M, fs_pipe=False, show_cigar=False, show_header=True):
        args = (self._h, qname.encode(), (qacc or "").encode(), M, int(fs_pipe), int(show_cigar), int(show_header))
        n = lib().bath_tophits_tabular_targets(*args, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib().bath_tophits_tabular_targets(*args, buf, n)
        return buf.raw[:n].decode()

    def annotations(self, M, fs_pipe=False):
        """Per reported hit, in rank order: the head of its 'Annotation for each hit' entry (p7_tophits_Domains)."""
        out = []
        for r in range(lib().bath_tophits_count(self._h)):
            n = lib().bath_tophits_domain_annotation(self._h, r, M, int(fs_pipe), None, 0)
            if n > 0:
                buf = C.create_string_buffer(n + 1)
                lib().bath_tophits_domain_annotation(self._h, r, M, int(fs_pipe), buf, n)
                out.append(buf.raw[:n].decode())
        return out

    def statistics(self, stats, params, nmodels, nnodes, nseqs):
        """The 'Internal pipeline statistics summary' block of the main output, without its timing lines (p7_pli_Statistics)."""
        args = (self._h, C.byref(stats), C.byref(params), nmodels, nnodes, nseqs)
        n = lib().bath_tophits_pipeline_statistics(*args, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib().bath_tophits_pipeline_statistics(*args, buf, n)
        return buf.raw[:n].decode()

    def targets(self, fs_pipe=False, textw=120):
        """The 'Scores for complete hits' block of bathsearch's main output (p7_tophits_Targets)."""
        n = lib().bath_tophits_targets(self._h, int(fs_pipe), textw, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib().bath_tophits_targets(self._h, int(fs_pipe), textw, buf, n)
        return buf.raw[:n].decode()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().bath_tophits_destroy(self._h)
            self._h = None


def FS3ForwardParser(ctx, om3, dna, logsum=LOGSUM_TABLE, want_xmx=False):
    """p7_ForwardParser_Frameshift_3Codons over a block of DNA windows."""
    return _fs3(lib().bath_hip_fs3_forward_parser, "fs3_forward_parser", ctx, om3, dna, logsum, want_xmx)


def FS3BackwardParser(ctx, om3, dna, logsum=LOGSUM_TABLE, want_xmx=False):
    """p7_BackwardParser_Frameshift_3Codons over a block of DNA windows."""
    return _fs3(lib().bath_hip_fs3_backward_parser, "fs3_backward_parser", ctx, om3, dna, logsum, want_xmx)


def _fs3(fn, what, ctx, om3, dna, logsum, want_xmx):
    sc = np.zeros(dna.n, dtype=np.float32)
    xmx, offs = None, None
    if want_xmx:
        offs = np.zeros(dna.n + 1, dtype=np.int64)
        np.cumsum((dna.lengths + 1) * 5, out=offs[1:])
        xmx = np.zeros(int(offs[-1]), dtype=np.float32)
    ctx._check(fn(ctx._h, om3._h, dna._h, logsum, _f32(sc), _f32(xmx), _i64(offs)), what)
    if want_xmx:
        return sc, [xmx[offs[i]:offs[i + 1]].reshape(-1, 5) for i in range(dna.n)]
    return sc


def FS5Envelopes(ctx, om5, dna, logsum=LOGSUM_TABLE, c5_compat=False, want_pp=False, want_oa=False):
    """Forward/Backward/Decoding/OptimalAccuracy/Null2 (frameshift, 5 codon lengths) per envelope."""
    M = om5.M
    res = (Fs5Result * max(dna.n, 1))()
    pp = ppo = oa = oao = None
    if want_pp:
        ppo = np.zeros(dna.n + 1, dtype=np.int64)
        np.cumsum((dna.lengths + 1) * (M + 1) * 8, out=ppo[1:])
        pp = np.zeros(int(ppo[-1]), dtype=np.float32)
    if want_oa:
        oao = np.zeros(dna.n + 1, dtype=np.int64)
        np.cumsum((dna.lengths + 1) * (M + 1) * 3, out=oao[1:])
        oa = np.zeros(int(oao[-1]), dtype=np.float32)
    ctx._check(lib().bath_hip_fs5_envelopes(ctx._h, om5._h, dna._h, logsum, 1 if c5_compat else 0, res,
                                            _f32(pp), _i64(ppo), _f32(oa), _i64(oao)), "fs5_envelopes")
    out = {"fwdsc": np.array([res[i].fwdsc for i in range(dna.n)], dtype=np.float32),
           "bcksc": np.array([res[i].bcksc for i in range(dna.n)], dtype=np.float32),
           "oasc": np.array([res[i].oasc for i in range(dna.n)], dtype=np.float32),
           "null2": np.array([list(res[i].null2) for i in range(dna.n)], dtype=np.float32).reshape(dna.n, KP)}
    if want_pp:
        out["pp"] = [pp[ppo[i]:ppo[i + 1]].reshape(-1, M + 1, 8) for i in range(dna.n)]
    if want_oa:
        out["oa"] = [oa[oao[i]:oao[i + 1]].reshape(-1, M + 1, 3) for i in range(dna.n)]
    return out
