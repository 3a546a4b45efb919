#!/bin/bash
# Kernel timeline of W worker contexts running strict --fs passes concurrently (kernels >= 0.3 ms), from a rocprofv3 kernel trace.
#   gpurun -- 'bash tools/fs_workers_timeline.sh 2'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
W=${1:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/fs_workers_timeline_$W
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 tools/fs_workers_probe.py --workers $W --passes 3 > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bath::", "")[:40], r.get("Queue_Id", r.get("Stream_Id", "?")), r.get("Workgroup_Size_X", "?"), r.get("Grid_Size_X", "?")) for r in rows]
ev.sort()
tiles = [i for i, e in enumerate(ev) if e[2].startswith("orf_tile")]
start_i = tiles[-(2 * $W * 2)]            # the last two passes of every worker: two parts each, two orf_tile launches per pass
t0 = ev[start_i][0]
end = max(e[1] for e in ev[start_i:])
print("last %d passes: %.2f ms" % (2 * $W, (end - t0) / 1e6))
# idle gaps: intervals > 0.5 ms with no kernel running
iv = sorted((s, e) for s, e, *_ in ev[start_i:])
cur = iv[0][1]
for s, e in iv[1:]:
    if s - cur > 500000: print("  GPU idle %8.3f -> %8.3f ms (%.2f)" % ((cur - t0) / 1e6, (s - t0) / 1e6, (s - cur) / 1e6))
    cur = max(cur, e)
for s, e, n, q, wg, grid in ev[start_i:]:
    if (e - s) > 300000:
        print("%8.3f -> %8.3f ms  (%6.3f)  q%-3s wg %-5s grid %-8s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, wg, grid, n))
PY
