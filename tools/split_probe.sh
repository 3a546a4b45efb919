#!/bin/bash
# A/B of the two parts' shares of a block (BATH_HIP_LANE_SPLIT): fresh process per setting, three processes each
for s in 0.5 0.6 0.65 0.7 0.75; do
  for r in 1 2 3; do
    BATH_HIP_LANE_SPLIT=$s python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fs --no-streamed 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $s run $r ms_per_step %.3f' % d['ms_per_step'])"
  done
done
