#!/bin/bash
# A/B of compile-time variants of the frameshift kernels on the GPU box: rebuilds bath_frameshift.o per variant.
#   gpurun -- 'bash tools/fs_variants.sh "-DBATH_FS_OA_WAVES=2" "-DBATH_FS_OA_WAVES=3" ...'
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -f bath_amd/csrc/bath_frameshift.o
  make -s -C bath_amd/csrc EXTRA="$v" 2>&1 | grep -E "error" | head -3
  echo "=== $v"
  bash tools/fs_kernels.sh 2>&1 | sed -n 2,10p | cut -c1-160
done
rm -f bath_amd/csrc/bath_frameshift.o
