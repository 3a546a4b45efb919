#!/bin/bash
# rocprofv3 evidence for profiles/r02_*: run on the GPU box through gpurun, e.g.
#   gpurun --timeout 2400 -- 'bash tools/prof_round2.sh'
# Pass 1: kernel trace + stats of the bench command itself (cascade + the --fs leg).  Pass 2: the same as one part (clean
# per-kernel durations).  PMC passes (never mixed with tracing), one counter set per run, on 200000-window blocks.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/round2
rm -rf $OUT; mkdir -p $OUT
BATH_HIP_TIMING=1 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_plain.json 2> $OUT/bench_stage_laps.txt
# 1a: the cascade alone, every ssv_orf_kernel launch a half-block launch of a step (its average is what roofline.kernel_ms of that run says)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats0 -o bench0 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fs --no-streamed --no-one-part > $OUT/bench_under_prof_cascade.log 2>&1
grep '^{"metric"' $OUT/bench_under_prof_cascade.log > $OUT/bench_under_prof_cascade.json
find $OUT/stats0 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_cascade.csv \;
rm -rf $OUT/stats0
# 1b: the whole default command (all legs)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_under_prof.log 2>&1
export BATH_HIP_LANES=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o bench1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fs --no-streamed > $OUT/bench_under_prof_1lane.log 2>&1
P="python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --windows 200000 --fs-windows 200000"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $P > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $P > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq1 -- $P > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- $P > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- $P > $OUT/pmc_grbm.log 2>&1
unset BATH_HIP_LANES
python3 tools/pmc_summary.py $OUT/pmc_by_kernel.json $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_grbm > $OUT/pmc_summary.txt
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats1 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_1lane.csv \;
rm -rf $OUT/stats $OUT/stats1 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_grbm
tail -1 $OUT/bench_under_prof.log | cut -c1-300
cat $OUT/pmc_summary.txt | head -40
