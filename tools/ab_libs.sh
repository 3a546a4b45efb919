#!/bin/bash
# A/B of two builds of the library on the same box, alternating: gpurun -- 'bash tools/ab_libs.sh bath_amd/libbathhip_prev.so'
cd $GRAFT_REPO_ROOT
OTHER=$GRAFT_REPO_ROOT/$1
for rep in 1 2 3; do
  for lib in "" "$OTHER"; do
    BATH_HIP_LIBRARY=$lib python3 bench.py --steps 30 --warmup 3 --no-fs --no-streamed --no-cpu-baseline --no-one-part 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-current}'.split('/')[-1], '%.3f ms/step' % d['ms_per_step'])"
  done
done
