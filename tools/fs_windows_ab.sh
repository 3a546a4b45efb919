#!/bin/bash
# A/B of the DNA-window stage: device (default) vs host (BATH_HIP_FS_WINDOWS_HOST=1), strict --fs passes on the bench block, stage laps.
#   gpurun -- 'bash tools/fs_windows_ab.sh'
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
for mode in device host device host; do
  if [ $mode = host ]; then export BATH_HIP_FS_WINDOWS_HOST=1; else unset BATH_HIP_FS_WINDOWS_HOST; fi
  python3 tools/fs_pass_laps.py 12 2> $OUT/laps_$mode.txt > /dev/null
  echo "== $mode"
  grep "^PASS" $OUT/laps_$mode.txt | tail -10 | awk '{s+=$3; n++} END {printf "mean of last %d passes: %.2f ms\n", n, s/n}'
  grep -E "fs:   (cascade|DNA windows|gather|branch)" $OUT/laps_$mode.txt | tail -5
done
