#!/bin/bash
# Kernel timeline of ONE query of the configs[3] slice (model index $1, default 5 = Thg1) against the 12.5 Mb genome: every kernel with
# start / end, and the gaps between kernels (host time).   gpurun -- 'bash tools/c4_query_timeline.sh 5'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
Q=${1:-5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/c4_query_timeline
rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<PY
import sys, os
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench, bath_amd as ba
from bath_amd import synth, dist as bdist
ctx = ba.Context(0)
hmms, g, planted = bench.c4_genome(ba, synth, int(12.5e6))
hmm = hmms[$Q]
om = ba.OProfile(ctx, ba.Profile(hmm)); pipe = ba.Pipeline(ctx, om, fs_pipe=False, ncbi_table=hmm.ct)
wins = bdist.split_targets([len(g)], hmm.max_length)
block = ba.SeqBlock(ctx, [g[s_:s_ + n] for _, s_, n, _ in wins]); block.set_context([c for _, _, _, c in wins])
for _ in range(4): pipe.run_hits(block)
ctx.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 $OUT/run.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bath::", "")[:44]) for r in rows)
tiles = [i for i, e in enumerate(ev) if e[2].startswith("orf_tile")]
s0 = tiles[-1]
t0 = ev[s0][0]
print("last call: %.3f ms of kernels span, %d kernels" % ((max(e[1] for e in ev[s0:]) - t0) / 1e6, len(ev) - s0))
cur = ev[s0][0]; busy = 0
for s, e, n in ev[s0:]:
    gap = (s - cur) / 1e3
    print("%8.3f -> %8.3f ms (%6.1f us)  gap before %7.1f us  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, gap if gap > 0 else 0.0, n))
    busy += max(0, e - max(s, cur)); cur = max(cur, e)
print("kernel-busy %.3f ms" % (busy / 1e6))
PY
