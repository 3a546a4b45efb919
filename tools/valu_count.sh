#!/bin/bash
# Dynamic instruction counts per kernel of the cascade (box-independent, unlike times): one --pmc pass on a 200000-window block.
#   gpurun -- 'bash tools/valu_count.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/valu_count
rm -rf $OUT; mkdir -p $OUT
export BATH_HIP_LANES=1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fs --no-streamed --no-one-part --windows 200000 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
tot = sum(v["SQ_INSTS_VALU"] for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:16]:
    d = max(n[k], 1)
    print("%-50s launches %3d  VALU %10.0f (%4.1f%%)  SALU %10.0f  LDS %9.0f per launch" % (k, d, v["SQ_INSTS_VALU"] / d, 100 * v["SQ_INSTS_VALU"] / tot, v["SQ_INSTS_SALU"] / d, v["SQ_INSTS_LDS"] / d))
PY
