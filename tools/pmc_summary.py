#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (one directory per pass, CSV output) into per-kernel averages.

usage: pmc_summary.py OUT.json PASS_DIR [PASS_DIR ...]
Every *counter_collection.csv under the pass directories is read; counters are averaged per kernel over its
dispatches.  Derived figures for the kernels of interest follow MI355X_MICROARCH.md: HBM bytes =
2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE, both reported in KB.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, ctrs in acc.items():
        e = {c: sum(v) / len(v) for c, v in ctrs.items()}
        e["dispatches"] = max(len(v) for v in ctrs.values())
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
        if "SQ_ACTIVE_INST_VALU" in e and "SQ_INSTS_VALU" in e and e["SQ_INSTS_VALU"] > 0:
            e["cycles_per_valu_inst"] = 4.0 * e["SQ_ACTIVE_INST_VALU"] / e["SQ_INSTS_VALU"]
        if "GRBM_GUI_ACTIVE" in e:
            cyc = e["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs: this is the kernel's duration in cycles
            e["kernel_cycles"] = cyc
            # SQ_ACTIVE_INST_VALU counts issued VALU instructions (4 cycles each) summed over the 1024 SIMDs;
            # SQ_LDS_IDX_ACTIVE counts LDS-array cycles summed over the 256 CUs
            if "SQ_ACTIVE_INST_VALU" in e:
                e["valu_busy_fraction"] = 4.0 * e["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc)
            if "SQ_LDS_IDX_ACTIVE" in e:
                e["lds_busy_fraction"] = e["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
        if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            e["lds_bank_conflict_fraction_of_lds_cycles"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
        res[k] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted(res, key=lambda k: -res[k].get("SQ_ACTIVE_INST_VALU", 0)):
        e = res[k]
        print("%-48s hbm=%s valu_busy=%s lds_busy=%s conflicts=%s" % (k[:48], "%.3g" % e["hbm_bytes_per_launch"] if "hbm_bytes_per_launch" in e else "-",
              "%.2f" % e["valu_busy_fraction"] if "valu_busy_fraction" in e else "-", "%.2f" % e["lds_busy_fraction"] if "lds_busy_fraction" in e else "-",
              "%.2f" % e["lds_bank_conflict_fraction_of_lds_cycles"] if "lds_bank_conflict_fraction_of_lds_cycles" in e else "-"))


if __name__ == "__main__":
    main()
