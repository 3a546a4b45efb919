#!/bin/bash
# A/B of the frameshift kernels' block size / register cap on the GPU box: rebuilds bath_frameshift.o per variant.
cd $GRAFT_REPO_ROOT
for v in "512 4" "384 3" "512 0" "256 2"; do
  set -- $v
  rm -f bath_amd/csrc/bath_frameshift.o
  make -s -C bath_amd/csrc EXTRA="-DBATH_FS_BLOCK=$1 -DBATH_FS_WAVES=$2" 2>&1 | grep -E "error" | head -3
  echo "=== block $1, waves/SIMD >= $2"
  bash tools/fs_kernels.sh 2>&1 | sed -n 2,9p | cut -c1-200
done
rm -f bath_amd/csrc/bath_frameshift.o
