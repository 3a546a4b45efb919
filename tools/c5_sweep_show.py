#!/usr/bin/env python3
"""configs[4] leg of the last bench run (gpurun_out/bench_detail.json): the pass, its chain kernels and the size sweep."""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_detail.json"))
c5 = d["c5"]
print("c5 ms_per_pass %.1f  kernels %s" % (c5["ms_per_pass"], json.dumps(c5["kernels_ms"])))
for e in c5["size_sweep"]:
    print(json.dumps(e))
print("parity all_equal:", c5.get("parity_check", {}).get("all_equal"))
