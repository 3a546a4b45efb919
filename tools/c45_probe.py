#!/usr/bin/env python3
"""bench.py's configs[3] / configs[4] legs on their own (no CPU parity sample): `python tools/c45_probe.py c5 [--c5-mb 125]`."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bath_amd as ba
from bath_amd import synth, dist as bdist

ap = argparse.ArgumentParser()
ap.add_argument("leg", choices=("c4", "c5", "both"))
ap.add_argument("--c4-mb", type=float, default=12.5)
ap.add_argument("--c5-mb", type=float, default=125.0)
ap.add_argument("--c45-sample-windows", type=int, default=6)
ap.add_argument("--c5-sample-windows", type=int, default=2)
args = ap.parse_args()
ctx = ba.Context(0)
out = {}
if args.leg in ("c4", "both"):
    out["c4"] = bench.c4_leg(ba, synth, bdist, ctx, args, None)
if args.leg in ("c5", "both"):
    out["c5"] = bench.c5_leg(ba, synth, bdist, ctx, args, None)
for k, v in out.items():
    v.pop("models", None)
    print(k, json.dumps(v, indent=1))
