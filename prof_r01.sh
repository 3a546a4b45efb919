set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_under_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --windows 200000 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --windows 200000 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq1 -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --windows 200000 > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --windows 200000 > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --windows 200000 > $OUT/pmc_grbm.log 2>&1
find $OUT -name "*.csv" | head -40
tail -2 $OUT/bench_under_prof.log | cut -c1-600
