#!/bin/bash
# A/B of the long models' Backward chain kernel: two waves per window (default) against one (BATH_HIP_FS_BWD_WPW=1), and the bit-identity
# tests that run long models.   gpurun -- 'bash tools/bwd_wpw_probe.sh'
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_fs_chain_gpu.py tests/test_fs_strict_gpu.py -x -q -m gpu -k "memory or long_model" 2>&1 | tail -3
for rep in 1 2; do
for w in 1 2; do
  echo "#### BATH_HIP_FS_BWD_WPW=$w"
  BATH_HIP_FS_BWD_WPW=$w timeout 900 python3 tools/chain_long_probe.py --n 1,327,1308,2616 2>&1 | grep backward
done
done
