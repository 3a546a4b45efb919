#!/bin/bash
# GPU-box probe of the --fs leg: frameshift parity tests, then the bench's fs object with stage timings (BATH_HIP_TIMING=1).
mkdir -p gpurun_out
python -m pytest tests/test_frameshift_gpu.py -x -q -m gpu > gpurun_out/fs_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/fs_tests.log
tail -15 gpurun_out/fs_tests.log
BATH_HIP_TIMING=1 timeout 900 python bench.py --no-cpu-baseline --no-streamed --no-concurrent --no-one-part --steps 2 --warmup 1 > gpurun_out/fs_probe.json 2> gpurun_out/fs_probe.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/fs_probe.json").read().strip().splitlines()[-1])
fs = d["fs"]
print("step ms", d["ms_per_step"], "fs ms_per_pass", fs["ms_per_pass"], "strict", fs["strict"]["ms_per_pass"], "domains", fs["domains"], fs["strict"])
for k, v in sorted(fs["kernels"].items(), key=lambda kv: -kv[1]["ms"]):
    print("%-28s %8.3f ms  launches %.1f" % (k, v["ms"], v["launches"]))
PY
grep "bath timing" gpurun_out/fs_probe.err | tail -60
