#!/bin/bash
# A/B of the chain kernels' loops (bath_fs_chain.hip): the default build (pollers, eight nodes per trip) against builds without the
# pollers (-DBATH_CHAIN_KEEPALIVE=0 -> tools/_ab/libbathhip_noka.so) and with 2 / 4 / 16 nodes per trip (-DBATH_CHAIN_UNROLL=..
# -> libbathhip_u2.so ...), and the clock build (-DBATH_CHAIN_CLOCK: ticks per chain node of block 0).
#   gpurun -- 'bash tools/keepalive_ab.sh [libs...]'
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
AB=$GRAFT_REPO_ROOT/tools/_ab
LIBS=${@:-default u2 noka u4 u16}
echo "#### parity (default build)"
timeout 1500 python3 -m pytest tests/test_fs_chain_gpu.py tests/test_frameshift_gpu.py tests/test_fs_strict_gpu.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for lib in $LIBS; do
  if [ $lib = default ]; then unset BATH_HIP_LIBRARY; else export BATH_HIP_LIBRARY=$AB/libbathhip_$lib.so; fi
  echo "#### $lib: configs[4]-like long windows (M = 1024)"
  timeout 600 python3 tools/chain_long_probe.py --n 1,327 2>&1 | grep -v "^bwd chain\|^fwd chain"
  echo "#### $lib: strict --fs passes on the bench block"
  timeout 600 python3 tools/fs_pass_laps.py 12 2> $OUT/ka_laps_$lib.txt > /dev/null
  grep "^PASS" $OUT/ka_laps_$lib.txt | tail -10 | awk '{s+=$3; n++} END {printf "mean of last %d passes: %.2f ms\n", n, s/n}'
  grep -E "fs: parsers|fs:   gather|single-domain regions" $OUT/ka_laps_$lib.txt | tail -3
done
done
export BATH_HIP_LIBRARY=$AB/libbathhip_clock.so
echo "#### clock: ticks per node, block 0 (M = 1024, 327 windows of 8 kb)"
timeout 600 python3 tools/chain_long_probe.py --n 327 2>&1 | grep "^bwd chain\|^fwd chain" | sort | uniq -c | sort -rn | head -8
