#!/bin/bash
# Speculative Backward (fs3_backward_spec): strict --fs passes on the bench block with the Backward parser of the K longest DNA windows
# running beside the Forward parser of all of them (BATH_HIP_FS_SPEC_K; 0 = none = rounds 1-5), and both parsers for ALL windows
# side by side (BATH_HIP_FS_SPEC_ALL=1) for comparison.     gpurun -- 'bash tools/fs_spec_probe.sh'
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
run() { echo "== $*"; env "$@" python3 tools/fs_pass_laps.py 10 2> $OUT/laps_spec.txt > /dev/null; grep "^PASS" $OUT/laps_spec.txt | tail -8 | awk '{s+=$3; n++} END {printf "mean of last %d passes: %.2f ms\n", n, s/n}'; grep -E "fs: parsers|fs: cascade \+ windows" $OUT/laps_spec.txt | tail -2; }
for k in 0 1024 2048 3072 4096 6144 0 3072; do run BATH_HIP_FS_SPEC_K=$k; done
run BATH_HIP_FS_SPEC_K=0 BATH_HIP_FS_SPEC_ALL=1 BATH_HIP_FS_SPEC_SHARE=1
