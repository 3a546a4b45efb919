#!/bin/bash
# Timeline of one strict --fs pass from a kernel trace: which kernels overlap (kernels >= 0.2 ms).
#   gpurun -- 'bash tools/fs_timeline.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fs_timeline
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 tools/fs_strict_probe.py --steps 3 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bath::", "")[:40], r.get("Queue_Id", r.get("Stream_Id", "?")), r.get("Workgroup_Size_X", "?"), r.get("Grid_Size_X", "?"), r.get("LDS_Block_Size", "?")) for r in rows]
ev.sort()
tiles = [i for i, e in enumerate(ev) if e[2].startswith("orf_tile")]
start_i = tiles[-2]                       # the last pass: two parts, two orf_tile launches
t0 = ev[start_i][0]
print("last pass: %.2f ms" % ((max(e[1] for e in ev[start_i:]) - t0) / 1e6))
for s, e, n, q, wg, grid, lds in ev[start_i:]:
    if (e - s) > 200000:
        print("%8.3f -> %8.3f ms  (%6.3f)  q%-3s wg %-5s grid %-8s lds %-7s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, wg, grid, lds, n))
PY
