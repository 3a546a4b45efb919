#!/usr/bin/env python3
"""Print the handful of numbers of a bench.py JSON line that the A/B runs compare.  usage: bench_show.py line.json [label]"""
import json, sys
o = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
lab = sys.argv[2] if len(sys.argv) > 2 else ""
r = lambda x: None if x is None else round(x, 2)
row = [lab, "step", r(o["ms_per_step"])]
if "c4" in o:
    row += ["c4", r(o["c4"]["ms_per_database_pass"]), r(o["c4"]["concurrent_queries"]["ms_per_database_pass"]), "full", r(o["c4"].get("full_job", {}).get("ms_per_database_pass"))]
if "fs" in o:
    f = o["fs"]
    row += ["fs", r(f["ms_per_pass"]), "x2", r((f.get("concurrent_blocks") or {}).get("ms_per_block")), "fast", r(f["fast"]["ms_per_pass"]), "frac", r(f["roofline"]["frac"] * 100)]
if "c5" in o:
    row += ["c5", r(o["c5"].get("ms_per_pass")), r(o["c5"].get("fast", {}).get("ms_per_pass"))]
if "concurrent_blocks" in o:
    row += ["cb", r(o["concurrent_blocks"]["ms_per_block"])]
print(*row)
