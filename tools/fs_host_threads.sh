#!/bin/bash
# What an N-rank --fs job needs from the host: strict passes on the bench block with the ensembles' host threads capped
# (BATH_HIP_HOST_THREADS = what a rank gets when N ranks share the node's cores), pass time and the ensembles' lap.
#   gpurun -- 'bash tools/fs_host_threads.sh'
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
echo "cores usable: $(nproc), cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for t in 64 16 8 4 2 1; do
  BATH_HIP_HOST_THREADS=$t python3 tools/fs_pass_laps.py 8 2> $OUT/laps_threads_$t.txt > /dev/null
  echo "== BATH_HIP_HOST_THREADS=$t"
  grep "^PASS" $OUT/laps_threads_$t.txt | tail -6 | awk '{s+=$3; n++} END {printf "mean of last %d passes: %.2f ms\n", n, s/n}'
  grep -E "ensemble threads, start to end|fs: ensembles \(host threads\)|clusters' envelope" $OUT/laps_threads_$t.txt | tail -3
done
