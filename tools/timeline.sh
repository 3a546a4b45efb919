#!/bin/bash
# Timeline of one cascade step from a kernel trace: which kernels of the two parts overlap, where the chip idles.
#   gpurun -- 'bash tools/timeline.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fs --no-streamed --no-one-part > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bath::", "")[:34], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows]
ev.sort()
# the steps: find the orf_tile launches; a step starts with two of them (one per part)
tiles = [i for i, e in enumerate(ev) if e[2].startswith("orf_tile")]
# take the 3rd-from-last pair as a timed step
start_i = tiles[-6] if len(tiles) >= 6 else tiles[0]
end_i = tiles[-4] if len(tiles) >= 6 else len(ev)
t0 = ev[start_i][0]
print("one step: %d kernels, %.2f ms" % (end_i - start_i, (max(e[1] for e in ev[start_i:end_i]) - t0) / 1e6))
for s, e, n, q in ev[start_i:end_i]:
    if (e - s) > 30000:
        print("%8.3f -> %8.3f ms  (%6.3f)  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
PY
