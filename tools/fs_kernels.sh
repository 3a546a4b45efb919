#!/bin/bash
# prints the bench's fs object compactly: gpurun -- 'bash tools/fs_kernels.sh'
cd $GRAFT_REPO_ROOT
BATH_HIP_TIMING=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streamed 2> gpurun_out/laps.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
f=d['fs']
print({k:f[k] for k in f if k not in ('kernels','workload','roofline','strict')})
for n,k in sorted(f['kernels'].items(), key=lambda x:-x[1]['ms']): print('  %-28s %7.2f ms  %4.1f launches  %8.1f Gcells/s  %7.1f GB/s'%(n,k['ms'],k['launches'],k['gcells_per_s'],k['algorithmic_GBps']))
print(f['roofline']['ms'], f['roofline']['frac'], f['strict']['ms_per_pass'], f['strict']['domains_identical_to_default_mode'], f['strict']['domains'])
"
grep "fs:" gpurun_out/laps.txt | tail -7
