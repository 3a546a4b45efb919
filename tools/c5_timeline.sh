#!/bin/bash
# Timeline of configs[4]'s strict --fs pass at a given genome size from a kernel trace (kernels >= 1 ms): what runs beside what.
#   gpurun -- 'bash tools/c5_timeline.sh 250'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
MB=${1:-250}
OUT=$GRAFT_REPO_ROOT/gpurun_out/c5_timeline
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 tools/c5_sweep_probe.py $MB > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bath::", "")[:44], r.get("Queue_Id", r.get("Stream_Id", "?")), r.get("Workgroup_Size_X", "?"), r.get("Grid_Size_X", "?")) for r in rows]
ev.sort()
chains = [i for i, e in enumerate(ev) if e[2].startswith("fs3_bwd_chain")]
last = chains[-1]
# the pass the last Backward chain belongs to: back to the previous long gap-free start (the pass's first orf_tile launch)
tiles = [i for i, e in enumerate(ev[:last]) if e[2].startswith("orf_tile")]
start_i = tiles[-1]
while start_i > 0 and ev[start_i][0] - ev[start_i - 1][1] < 2_000_000 and not ev[start_i - 1][2].startswith("fs5"): start_i -= 1
t0 = ev[start_i][0]
end = max(e[1] for e in ev[start_i:])
print("last pass: %.1f ms of kernels start to end" % ((end - t0) / 1e6))
for s, e, n, q, wg, grid in ev[start_i:]:
    if (e - s) > 1_000_000:
        print("%9.2f -> %9.2f ms  (%8.2f)  q%-3s wg %-5s grid %-8s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, wg, grid, n))
PY
