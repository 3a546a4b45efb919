"""CPU tier: the C-ABI library loads and exports every symbol include/bath_hip.h declares, the host-side
model code (bhmm reader, p7_ProfileConfig[_fs]) is bit-identical to the oracle's independent restatement,
and the backend fails loudly without a GPU (no CPU fallback)."""
import re

import numpy as np
import pytest

import bath_amd as ba
import oracle_lib as ol


def test_header_symbols_are_exported():
    hdr = open(ba._ROOT + "/include/bath_hip.h").read()
    declared = set(re.findall(r"\b(bath_[a-zA-Z0-9_]+)\s*\(", hdr))
    declared -= {"bath_hip_ctx", "bath_hip_oprofile", "bath_hip_fsprofile", "bath_hip_seqs"}
    assert declared == set(ba.ABI), (declared ^ set(ba.ABI))
    lib = ba.lib()                                   # binds every name; raises AttributeError on a missing export
    for name in declared:
        assert getattr(lib, name) is not None


@pytest.mark.parametrize("name", ["Caudal_act.bhmm", "PTH2.bhmm", "MET-ct4.bhmm"])
def test_profile_config_matches_oracle(name):
    path = ol.GOLDEN + "/" + name
    hmm = ba.HMM(path)
    m = ol.Model(path)
    assert (hmm.M, hmm.ct) == (m.M, m.hmm.contents.ct)
    assert np.array_equal(hmm.evparam, np.array(m.hmm.contents.evparam[:], np.float32))
    tsc, rsc, xsc = ba.Profile(hmm, 100).arrays()
    g = m.gm.contents
    otsc = np.ctypeslib.as_array(g.tsc, shape=(m.M + 1, 8))[: m.M]
    orsc = np.ctypeslib.as_array(g.rsc, shape=(ol.KP, m.M + 1, 2))
    oxsc = np.array([[g.xsc[i][j] for j in range(2)] for i in range(4)], np.float32)
    assert np.array_equal(tsc.view(np.uint32), otsc.view(np.uint32))
    assert np.array_equal(rsc.view(np.uint32), orsc.view(np.uint32))
    assert np.array_equal(xsc.view(np.uint32), oxsc.view(np.uint32))


@pytest.mark.parametrize("codon_lengths", [3, 5])
def test_fs_profile_config_matches_oracle(codon_lengths):
    path = ol.GOLDEN + "/Caudal_act.bhmm"
    hmm = ba.HMM(path)
    m = ol.Model(path)
    fs = ba.FSProfile(hmm, codon_lengths, 100)
    tsc, rsc, codons, indel = fs.arrays()
    o = m.fs(codon_lengths).contents
    nrows = o.maxcodons + ol.KP
    assert fs.maxcodons == o.maxcodons
    orsc = np.ctypeslib.as_array(o.rsc, shape=(nrows, m.M + 1))
    assert np.array_equal(rsc.view(np.uint32), orsc.view(np.uint32))
    ocod = np.ctypeslib.as_array(o.codons, shape=(m.M + 1, o.maxcodons))
    oind = np.ctypeslib.as_array(o.indel_pos, shape=(m.M + 1, o.maxcodons))
    assert np.array_equal(codons[1:], ocod[1:]) and np.array_equal(indel[1:], oind[1:])
    # quasi-codons whose every reading is a stop codon (e.g. TATAA) legitimately stay at -inf (modelconfig.c:466-490)
    assert np.isfinite(rsc[: o.maxcodons, 1:]).mean() > 0.99


def test_gencode_tables():
    b1, b4 = ba.gencode_basic(1), ba.gencode_basic(4)
    tga = 16 * 3 + 4 * 2 + 0
    assert b1[tga] == 27 and b4[tga] == ba.AMINO_SYMS.index("W")
    o = np.zeros(64, np.uint8); ol.lib().bo_gencode_basic(4, ol.u8(o))
    assert np.array_equal(o, b4)
    with pytest.raises(ba.BathError):
        ba.gencode_basic(7)


def test_no_gpu_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ba.BathError, match="no CPU fallback"):
        ba.Context(0)


def test_header_is_plain_c_and_links(tmp_path):
    """include/bath_hip.h is the boundary a C host (BATH itself) compiles against: it must be valid C99 with no C++ or torch
    types, and a C program using it must link against libbathhip.so (host-side entry points only: no GPU here)."""
    import os
    import shutil
    import subprocess
    ROOT = ba._ROOT
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "host.c"
    src.write_text(
        '#include <stdio.h>\n#include "bath_hip.h"\n'
        "int main(int argc, char **argv) {\n"
        "  bath_hmm *hmm = NULL; bath_profile *gm = NULL; bath_pipeline_params prm;\n"
        "  if (argc < 2 || bath_hmmfile_read(argv[1], 0, &hmm) != BATH_OK) return 2;\n"
        "  if (bath_profile_config(hmm, 100, &gm) != BATH_OK) return 3;\n"
        "  bath_pipeline_params_default(&prm, 1);\n"
        "  bath_tophits *th = bath_tophits_create();\n"
        "  if (bath_tophits_finalize(th, 1000, hmm->max_length, 10.0) != BATH_OK) return 4;\n"
        '  printf("%s %d %s %d %g %ld\\n", hmm->name, hmm->M, hmm->acc, gm->M, prm.F4, (long) bath_tophits_count(th));\n'
        "  bath_tophits_destroy(th); bath_profile_destroy(gm); bath_hmm_destroy(hmm);\n"
        "  return 0;\n}\n")
    exe = tmp_path / "host"
    libdir = os.path.dirname(ba.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lbathhip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe), os.path.join(ROOT, "tests", "golden", "PTH2.bhmm")], text=True).split()
    assert out == ["PTH2", "116", "PF01981.11", "116", "0.0005", "0"]
