#!/bin/bash
# Kernel trace of N strict --fs passes: per pass, start / end (relative to the pass's first kernel) and duration of the long kernels --
# which kernel is longer in the passes that take 10-15 ms more than the median.   gpurun -- 'bash tools/fs_passes_trace.sh 30'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
N=${1:-30}
OUT=$GRAFT_REPO_ROOT/gpurun_out/fs_passes_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 tools/fs_pass_times.py $N > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-200
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bath::", "")[:28]) for r in rows)
tiles = [i for i, e in enumerate(ev) if e[2].startswith("orf_tile")]
starts = tiles[0::2]                       # two parts per pass
names = ["fs3_fwd_chain_half_kernel<5>", "fs3_bwd_chain_half_kernel<5>", "fs5_fwd_chain_kernel<3, 256>", "fs5_fwd_wf_kernel<false, fal", "fs5_bwd_wf_kernel<false, fal", "fs5_decode_oa_mw_kernel<2>", "fs5_trace_kernel"]
print("pass   span | " + " | ".join(n[:14].ljust(20) for n in names))
for p, s_i in enumerate(starts):
    e_i = starts[p + 1] if p + 1 < len(starts) else len(ev)
    seg = ev[s_i:e_i]
    t0 = seg[0][0]
    span = (max(e[1] for e in seg) - t0) / 1e6
    cells = []
    for n in names:
        k = [e for e in seg if e[2] == n[:28]]
        if not k: cells.append("-".ljust(20)); continue
        k0 = k[0]
        cells.append(("%5.1f-%5.1f (%4.1f)" % ((k0[0] - t0) / 1e6, (k0[1] - t0) / 1e6, (k0[1] - k0[0]) / 1e6)).ljust(20))
    print("%3d %7.1f | %s" % (p, span, " | ".join(cells)))
PY
