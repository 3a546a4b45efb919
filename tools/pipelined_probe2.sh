#!/bin/bash
# non-persistent SSV x tail-stream priorities x lanes, single call and back-to-back (tools/pipelined_probe.py)
cd $GRAFT_REPO_ROOT
for cfg in "0 0" "16 0" "16 1" "4 1" "64 1"; do
  set -- $cfg
  echo "=== BATH_HIP_SSV_CHUNK=$1 BATH_HIP_TAIL_PRIO=$2"
  BATH_HIP_SSV_CHUNK=$1 BATH_HIP_TAIL_PRIO=$2 python3 tools/pipelined_probe.py 2>&1 | grep -v "reps  1" | tail -6
done
