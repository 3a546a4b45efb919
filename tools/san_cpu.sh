#!/bin/bash
# Sanitizers on the CPU build (the reference ships --enable-asan / --enable-tsan, configure.ac:211,300-311): builds libbathhip_san.so
# (make SAN=1: ASan + UBSan on the HOST side of every translation unit; no GPU sanitizer exists on this pool) and runs the CPU-tier
# tests that reach the host code -- model reader and profile construction (host_model.cpp), hit list and --tblout (bath_tophits.hip),
# hit streams and work division (bath_dist.hip), alignment blocks (bath_alidisplay.hip), ensemble clustering self-tests
# (bath_ensemble.hip), option plumbing, the impl_hip harness build, the gloo exchange -- against it.
# Usage: tools/san_cpu.sh [output file]     (default profiles/r06_sanitizer_cpu.txt)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/profiles/r06_sanitizer_cpu.txt}
ASAN_RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
make -s -j6 -C "$ROOT/bath_amd/csrc" SAN=1 || exit 1
export BATH_HIP_LIBRARY=$ROOT/bath_amd/libbathhip_san.so
export LD_PRELOAD=$ASAN_RT
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
cd "$ROOT"
{
  echo "# tools/san_cpu.sh  $(date -u +%Y-%m-%dT%H:%MZ)  hipcc -Xarch_host -fsanitize=address,undefined (host side of all 17 translation units), runtime $ASAN_RT"
  echo "# 1. the instrumentation is live: a caller that overstates nbytes makes bath_hits_deserialize read past a heap block; ASan must stop it"
  python3 - <<'PY' 2>&1 | grep -E "library mapped|ERROR: AddressSanitizer|#1 .*bath_hits_deserialize|NOT REPORTED" | sed -e 's/ at pc.*//' | head -5
import ctypes as C, sys
import bath_amd as ba
L = ba.lib()
print("library mapped:", [l.split()[-1] for l in open("/proc/self/maps") if "libbathhip" in l][0])
sys.stdout.flush()
d = ba.FsDomain(); d.reported = 1; d.cigar = "30M"
b = bytearray(ba.HitArray.from_domains([d]).to_bytes())
b[-1] = ord("M")                                              # the CIGAR loses its terminating NUL ...
libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]
p = libc.malloc(len(b)); C.memmove(p, bytes(b), len(b))      # ... in a heap block of exactly the stream's size ...
H = C.c_void_p()
L.bath_hits_deserialize(C.c_void_p(p), len(b) + 64, C.byref(H))   # ... and the caller lies about nbytes: the scan for the NUL leaves the block
print("NOT REPORTED")
PY
  echo "# 2. the CPU-tier tests that reach host code, against the instrumented library"
  python3 -m pytest tests/test_abi_cpu.py tests/test_tophits_cpu.py tests/test_multi_gpu_c_cpu.py tests/test_ensemble_cpu.py tests/test_alidisplay_cpu.py \
      tests/test_options_cpu.py tests/test_easel_pieces_cpu.py tests/test_dist_cpu.py tests/test_sse_cpu.py -q -p no:cacheprovider 2>&1 | tail -15
} > "$OUT" 2>&1
cat "$OUT"
