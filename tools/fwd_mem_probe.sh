cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_fs_chain_gpu.py -x -q -m gpu -k "memory" 2>&1 | tail -15
for mem in 0 1; do
  echo "#### BATH_HIP_FS_FWD_MEM=$mem"
  BATH_HIP_FS_FWD_MEM=$mem timeout 900 python3 tools/chain_long_probe.py --n 327,1308,2616 2>&1 | grep forward
done
echo "#### clock build, BATH_HIP_FS_FWD_MEM=1"
BATH_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/_ab/libbathhip_clock.so BATH_HIP_FS_FWD_MEM=1 timeout 600 python3 tools/chain_long_probe.py --n 1308 2>&1 | grep "fwd mem" | sort | uniq -c | sort -rn | head -4
