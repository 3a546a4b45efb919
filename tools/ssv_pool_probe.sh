#!/bin/bash
# Item "SSV residue pool in work-list order": what the SSV kernel ITSELF would gain from a pool with no over-fetch, measured before
# building the pool.  Rebuilds bath_pipeline.o with -DBATH_SSV_POOL_PROBE (the kernel then reads a wave's residues as 512 contiguous
# bytes per load out of the first 128 MB of the amino-acid pool: WRONG results, the traffic of a pool without over-fetch) on the GPU box
# and times the kernel both ways (HIP events, roofline.kernel_ms).  Every command under its own timeout.
#   gpurun --timeout 1500 -- 'bash tools/ssv_pool_probe.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ssv_pool_probe
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fs --no-streamed --no-concurrent --no-c45"
show() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']
print(sys.argv[2], 'ms_per_step', d['ms_per_step'], 'ssv_orf kernel_ms (half block)', r['kernel_ms'], 'valu_frac', r.get('valu_frac'))" $1 $2; }
for v in base probe; do
  if [ $v = probe ]; then rm -f bath_amd/csrc/bath_pipeline.o; make -s -C bath_amd/csrc EXTRA=-DBATH_SSV_POOL_PROBE 2>&1 | grep -E "error" | head -3; fi
  timeout 300 $B 2> /dev/null | tail -1 > $OUT/bench_$v.json; show $OUT/bench_$v.json $v
done
