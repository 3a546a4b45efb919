"""Builds the impl_hip/ shim layer against the test harness header (tests/impl_hip_harness) into a shared object that the
impl_hip tests drive through ctypes.  TEST INFRASTRUCTURE."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "impl_hip_harness", "libimplhip_test.so")
SRCS = [os.path.join(ROOT, "impl_hip", "impl_hip_objects.c"), os.path.join(ROOT, "impl_hip", "impl_hip_kernels.c"),
        os.path.join(HERE, "impl_hip_harness", "harness.c")]
DEPS = SRCS + [os.path.join(ROOT, "impl_hip", "impl_hip.h"), os.path.join(HERE, "impl_hip_harness", "hmmer.h"), os.path.join(ROOT, "include", "bath_hip.h")]


def build():
    import bath_amd
    lib = bath_amd.build()
    if (not os.path.exists(SO)) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in DEPS + [lib]):
        subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-Werror", "-Wno-unused-function", "-fPIC", "-shared",
                               "-I" + os.path.join(HERE, "impl_hip_harness"), "-I" + os.path.join(ROOT, "impl_hip"), "-I" + os.path.join(ROOT, "include")]
                              + SRCS + ["-o", SO, "-L" + os.path.dirname(lib), "-lbathhip", "-Wl,-rpath," + os.path.dirname(lib), "-lm"])
    return SO
