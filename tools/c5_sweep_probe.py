#!/usr/bin/env python3
"""bench.py's c5.size_sweep on its own: `python tools/c5_sweep_probe.py 125,1000`."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bath_amd as ba
from bath_amd import synth, dist as bdist
sizes = [float(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "125,1000").split(",")]
ctx = ba.Context(0)
for row in bench.c5_size_sweep(ba, synth, bdist, ctx, sizes):
    print(json.dumps(row))
