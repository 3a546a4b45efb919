"""impl_hip/ -- the drop-in for the reference's impl_sse/ directory -- checked on the CPU tier:

* every function impl_hip.h declares has EXACTLY the prototype impl_sse.h declares (name, return type, argument types and
  names), read from the reference tree when it is present (the build container; skipped on the GPU box);
* every impl_sse symbol that the bathsearch path's generic code references (p7_pipeline.c, p7_domaindef.c, bathsearch.c,
  p7_scoredata.c, p7_alidisplay.c) is declared by impl_hip.h;
* impl_hip/ *.c compile warning-free against the test harness header and define every declared function (no stubs left
  undefined: the shared object exports each one)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
IMPL_HIP_H = os.path.join(ROOT, "impl_hip", "impl_hip.h")


def prototypes(path):
    """name -> normalised 'extern' prototype text"""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"extern\s+([^;{}]*?\))\s*;", text, flags=re.S):
        proto = re.sub(r"\s+", " ", m.group(1)).strip()
        proto = re.sub(r"\s*([(),*])\s*", r"\1", proto)
        name = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\(", proto).group(1)
        out[name] = proto
    return out


def test_declared_functions_are_defined():
    import impl_hip_build
    so = impl_hip_build.build()
    syms = subprocess.run(["nm", "-D", "--defined-only", so], stdout=subprocess.PIPE, text=True, check=True).stdout
    defined = {l.split()[-1] for l in syms.splitlines() if " T " in l}
    declared = {n for n in prototypes(IMPL_HIP_H)}
    assert len(declared) >= 55
    assert declared <= defined, sorted(declared - defined)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_prototypes_equal_impl_sse():
    want = prototypes(os.path.join(REF, "impl_sse", "impl_sse.h"))
    got = prototypes(IMPL_HIP_H)
    ours = {"impl_hip_init", "impl_hip_context"}
    for name, proto in got.items():
        if name in ours:
            continue
        assert name in want, name + " is not an impl_sse function"
        assert proto == want[name], (proto, want[name])


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_every_symbol_the_path_references_is_declared():
    sse = open(os.path.join(REF, "impl_sse", "impl_sse.h")).read()
    impl_functions = set(prototypes(os.path.join(REF, "impl_sse", "impl_sse.h"))) | {"impl_Init", "p7_oprofile_FGetEmission"}
    hip = open(IMPL_HIP_H).read()
    declared = set(prototypes(IMPL_HIP_H)) | set(re.findall(r"^(p7_[A-Za-z0-9_]+|impl_Init)\(", hip, flags=re.M))
    needed = set()
    for f in ("p7_pipeline.c", "p7_domaindef.c", "bathsearch.c", "p7_scoredata.c", "p7_alidisplay.c"):
        body = open(os.path.join(REF, f)).read()
        body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
        needed |= {s for s in re.findall(r"\b(p7_[A-Za-z0-9_]+|impl_Init)\s*\(", body) if s in impl_functions}
    assert len(needed) >= 50
    assert needed <= declared, sorted(needed - declared)
    # ... and the four types hmmer.h embeds by name
    for t in ("P7_OPROFILE", "P7_FS_OPROFILE", "P7_OIVX", "P7_OMX"):
        assert re.search(r"}\s*" + t + r"\s*;", hip) and re.search(r"}\s*" + t + r"\s*;", sse)


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "hmmer.h")), reason="the reference tree is only present in the build container")
def test_impl_hip_compiles_against_the_reference_hmmer_h(tmp_path):
    """The drop-in compiles WHERE THE REFERENCE INCLUDES IT: impl_hip/*.c through the reference's real src/hmmer.h (its struct
    layouts, its prototypes for every generic function impl_hip calls, its section 14 including the implementation header,
    hmmer.h:1044-1052), `gcc -fsyntax-only -Wall -Werror`.  easel is absent from the image, so the headers hmmer.h:38-52 names are
    the self-written stand-ins of tests/ref_compile_stubs (opaque types, a handful of prototypes); no object code is produced and
    nothing of the reference is built.  hmmer.h is reached through a symbolic link in a scratch directory so that its
    `#include "impl_sse/impl_sse.h"` resolves, relative to the link, to a one-line redirect to impl_hip.h -- what an `--enable-hip`
    branch of hmmer.h:1044-1052 would include."""
    stubs = os.path.join(ROOT, "tests", "ref_compile_stubs")
    os.symlink(os.path.join(REF, "hmmer.h"), tmp_path / "hmmer.h")
    (tmp_path / "impl_sse").mkdir()
    (tmp_path / "impl_sse" / "impl_sse.h").write_text(open(os.path.join(stubs, "impl_sse", "impl_sse.h")).read())
    srcs = [os.path.join(ROOT, "impl_hip", f) for f in sorted(os.listdir(os.path.join(ROOT, "impl_hip"))) if f.endswith(".c")]
    assert len(srcs) == 2
    cmd = ["gcc", "-std=gnu11", "-fsyntax-only", "-Wall", "-Werror", "-I" + str(tmp_path), "-I" + stubs,
           "-I" + os.path.join(ROOT, "impl_hip"), "-I" + os.path.join(ROOT, "include")] + srcs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    # the check has teeth: the translation units really went through the reference's header and the redirect (-H lists what was read)
    r = subprocess.run(cmd + ["-H"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    seen = r.stdout
    assert str(tmp_path / "hmmer.h") in seen and os.path.join("impl_hip", "impl_hip.h") in seen and "/root/reference/src/impl_sse/impl_sse.h" not in seen
    # ... and an undeclared helper is an error here (what round 5's esl_abc_FAvgScVec was)
    bad = tmp_path / "bad.c"
    bad.write_text('#include "hmmer.h"\nint f(const ESL_ALPHABET *a, float *v) { return esl_abc_NoSuchHelper(a, v); }\n')
    r = subprocess.run(cmd[:-2] + [str(bad)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 and "esl_abc_NoSuchHelper" in r.stdout
