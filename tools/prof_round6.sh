#!/bin/bash
# rocprofv3 evidence for profiles/r06_*: run on the GPU box through gpurun, e.g.
#   gpurun --timeout 2400 -- 'bash tools/prof_round6.sh'
# Tracing and PMC collection are always separate runs.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/round6
rm -rf $OUT; mkdir -p $OUT
# 0: the bench line itself (plain: no laps, no profiler), then the stage laps of the cascade and the --fs pass from a second, shorter run
#    (BATH_HIP_TIMING=1 adds a host synchronisation per lap: its passes are a few ms longer than the plain ones)
python3 bench.py --steps 5 --warmup 1 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
BATH_HIP_TIMING=1 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-c45 --no-streamed --no-concurrent --no-one-part > /dev/null 2> $OUT/bench_stage_laps.txt
# 1a: the cascade alone: every ssv_orf_kernel launch is a half-block launch of a timed step
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats0 -o bench0 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fs --no-streamed --no-concurrent --no-one-part --no-c45 > $OUT/bench_under_prof_cascade.log 2>&1
grep '^{"metric"' $OUT/bench_under_prof_cascade.log | tail -1 > $OUT/bench_under_prof_cascade.json   # stdout's compact line is the last one (stderr carries the full record)
find $OUT/stats0 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_cascade.csv \;
rm -rf $OUT/stats0
# 1b: the whole default command (all legs: cascade, streamed, --fs strict + fast, c4, c5)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_under_prof.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/stats
# 1c: the --fs pass alone, strict mode (the library's default) and fast mode
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o fs -- python3 tools/fs_strict_probe.py --steps 3 > $OUT/fs_strict_under_prof.log 2>&1
find $OUT/stats2 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_fs_strict.csv \;
rm -rf $OUT/stats2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3 -o fs -- python3 tools/fs_strict_probe.py --steps 3 --strict 0 > $OUT/fs_fast_under_prof.log 2>&1
find $OUT/stats3 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_fs_fast.csv \;
rm -rf $OUT/stats3
# 1d: configs[3] / configs[4] legs alone (a small main block keeps the rest of the command short)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats4 -o c45 -- python3 bench.py --steps 2 --warmup 1 --windows 20000 --no-cpu-baseline --no-fs --no-streamed --no-concurrent --no-one-part > $OUT/c45_under_prof.log 2>&1
find $OUT/stats4 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_c4_c5.csv \;
rm -rf $OUT/stats4
# 2: PMC passes, one counter set per run, one lane, on 200000-window blocks (both legs; the --fs leg runs strict then fast)
export BATH_HIP_LANES=1
P="python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-streamed --no-concurrent --no-one-part --no-c45 --windows 200000 --fs-windows 200000"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $P > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $P > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq1 -- $P > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- $P > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- $P > $OUT/pmc_grbm.log 2>&1
unset BATH_HIP_LANES
# 3: two worker contexts running strict --fs passes at the same time: the kernel timeline (profiles/r06_fs_concurrent_timeline.txt)
bash tools/fs_workers_timeline.sh 2 > $OUT/fs_concurrent_timeline.txt 2>&1
bash tools/fs_timeline.sh > $OUT/fs_pass_timeline.txt 2>&1
python3 tools/fs_workers_probe.py --workers 1,2,3 --passes 6 > $OUT/fs_workers.txt 2>&1
# 4: configs[3]: one query's kernel timeline (the longest model and a typical one), the job item by item and with its workers
bash tools/c4_query_timeline.sh 3 > $OUT/c4_query_timeline_RtcB.txt 2>&1
bash tools/c4_query_timeline.sh 2 > $OUT/c4_query_timeline_PTH2.txt 2>&1
python3 tools/c4_items_probe.py 100 6 2>&1 | grep -v amdgpu.ids > $OUT/c4_items.txt
# 5: this round's A/Bs: the DNA-window stage on the device against the host path; the ensembles under capped host threads; the N-rank
#    legs with real kernels on this one GPU (gloo, shared device) at 2 and 3 ranks
bash tools/fs_windows_ab.sh > $OUT/fs_windows_ab.txt 2>&1
bash tools/fs_host_threads.sh > $OUT/fs_host_threads.txt 2>&1
for n in 2 3; do
  BATH_BENCH_BACKEND=gloo BATH_BENCH_SHARE_DEVICE=1 python3 bench.py --gpus $n --steps 2 --warmup 1 --scaling strong --windows 20000 --fs-windows 20000 --c4-total-mb 12 --c5-total-mb 30 --no-cpu-baseline > $OUT/nrank_${n}_line.json 2> $OUT/nrank_${n}.err
  cp gpurun_out/bench_detail.json $OUT/nrank_${n}_detail.json
done
python3 tools/chain_long_probe.py > $OUT/chain_long_probe.txt 2>&1
# 6: the chain kernels of long models: the Forward kernel with the rows' history in memory against the register kernel; configs[4]'s
#    pass at 250 Mb kernel by kernel; the whole GPU tier
bash tools/fwd_mem_probe.sh > $OUT/fwd_mem_probe.txt 2>&1
bash tools/c5_timeline.sh 250 > $OUT/c5_timeline_250mb.txt 2>&1
python3 -m pytest tests -m gpu -x -q > $OUT/gputest_all.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_by_kernel.json $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_grbm > $OUT/pmc_summary.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_grbm
tail -1 $OUT/bench_under_prof.log | cut -c1-300
head -60 $OUT/pmc_summary.txt
