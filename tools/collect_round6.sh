#!/bin/bash
# gpurun_out/round6 (tools/prof_round6.sh) -> profiles/r06_*
cd "$(dirname "$0")/.."
S=gpurun_out/round6; D=profiles
tail -1 $S/bench_plain.json > $D/r06_bench_line_plain.json
grep '^{"metric"' $S/bench_under_prof.log | tail -1 > $D/r06_bench_line.json
cp $S/bench_under_prof_cascade.json $D/r06_bench_cascade_under_prof.json
grep "bath timing" $S/bench_stage_laps.txt > $D/r06_bench_stage_laps.txt
cp $S/kernel_stats.csv $D/r06_bench_kernel_stats.csv
cp $S/kernel_stats_cascade.csv $D/r06_bench_cascade_kernel_stats.csv
cp $S/kernel_stats_fs_strict.csv $D/r06_fs_strict_kernel_stats.csv
cp $S/kernel_stats_fs_fast.csv $D/r06_fs_fast_kernel_stats.csv
cp $S/kernel_stats_c4_c5.csv $D/r06_c4_c5_kernel_stats.csv
cp $S/gputest_all.log $D/r06_gputest_all.log
for f in fs_pass_timeline fs_concurrent_timeline fs_workers c4_query_timeline_RtcB c4_query_timeline_PTH2 c4_items chain_long_probe fs_windows_ab fs_host_threads pmc_summary fwd_mem_probe c5_timeline_250mb bwd_wpw_probe; do
  grep -v "amdgpu.ids\|simple_timer" $S/$f.txt > $D/r06_$f.txt
done
cp $S/pmc_by_kernel.json $D/r06_pmc_by_kernel_200k_windows.json
python3 tools/pmc_extract.py $S/pmc_by_kernel.json $D r06
python3 - <<'PY'
import json
for n in (2, 3):
    d = json.load(open("gpurun_out/round6/nrank_%d_detail.json" % n))
    keep = {k: d[k] for k in ("metric", "n_gpus", "scaling", "value", "ms_per_step", "residues_per_step", "survivors", "hits_gathered", "strong_scaling_check", "c4", "c5", "fs")}
    keep["how"] = ("BATH_BENCH_BACKEND=gloo BATH_BENCH_SHARE_DEVICE=1 python3 bench.py --gpus %d --steps 2 --warmup 1 --scaling strong --windows 20000 --fs-windows 20000 "
                   "--c4-total-mb 12 --c5-total-mb 30 --no-cpu-baseline (what tests/test_nrank_gpu.py runs): the ranks share GPU 0, real kernels, collectives over gloo; "
                   "the timings are NOT scaling figures (one device)" % n)
    json.dump(keep, open("profiles/r06_nrank_%dranks_one_gpu.json" % n, "w"), indent=1)
PY
ls $D | grep r06
