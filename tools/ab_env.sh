#!/bin/bash
# A/B of environment settings on the same box, alternating: gpurun -- 'bash tools/ab_env.sh "A=1 B=2" "A=3"'
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for cfg in "$@"; do
    env $cfg python3 bench.py --steps 30 --warmup 3 --no-fs --no-streamed --no-cpu-baseline --no-one-part 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %.3f ms/step' % ('$cfg', d['ms_per_step']))"
  done
done
