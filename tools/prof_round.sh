#!/bin/bash
# rocprofv3 evidence for profiles/: run on the GPU box through gpurun, e.g.
#   gpurun --timeout 1500 -- 'bash tools/prof_round.sh'
# Pass 1: kernel trace + stats of the bench command itself.  Passes 2-6: PMC counters, one set per run (never mixed
# with tracing), on a 200000-window block (1/5 of the bench block) to keep the serialised PMC passes short.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_under_prof.log 2>&1
export BATH_HIP_LANES=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o bench1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_under_prof_1lane.log 2>&1
unset BATH_HIP_LANES
P="python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --windows 200000"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $P > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $P > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq1 -- $P > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- $P > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- $P > $OUT/pmc_grbm.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_by_kernel.json $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_grbm
cp $OUT/stats/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats1 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_1lane.csv \;
tail -1 $OUT/bench_under_prof.log | cut -c1-400
