"""profiles/r02_fs_pmc.json and profiles/r02_ssv_orf_pmc.json from the per-kernel counter summary tools/prof_round2.sh leaves in
gpurun_out/round2/pmc_by_kernel.json.  Usage: python tools/pmc_extract.py gpurun_out/round2/pmc_by_kernel.json profiles"""
import json, sys

src, dst = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r02"          # file prefix: the round the counters were collected in
by = json.load(open(src))
kernels = by.get("kernels", by)
fs = {k: v for k, v in kernels.items() if any(t in k for t in ("fs3_", "fs5_", "fs_bwd", "fs_regions", "fs_bias", "fs_window"))}
if rnd != "r02":
    json.dump({"_what": "rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; two SQ sets; GRBM -- one counter set per run, never mixed with tracing) of the bench command on "
                        "200000-window blocks, BATH_HIP_LANES=1, averaged per kernel over its dispatches (tools/prof_round3.sh, tools/pmc_summary.py): the --fs leg runs "
                        "in strict mode (fs3_fwd_chain*, fs3_bwd_chain, fs5_fwd_chain, fs5_fwd_wf, fs5_bwd_wf, fs5_bwd_x) and then in fast mode (<., 0> instantiations). "
                        "hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB as MI355X_MICROARCH.md prescribes for gfx950.",
               "kernels": fs}, open(dst + "/%s_fs_pmc.json" % rnd, "w"), indent=1, sort_keys=True)
    fs = None
if fs is not None:
  json.dump({
    "_what": "rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; two SQ sets; GRBM -- one counter set per run, never mixed with tracing) of "
             "`python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --windows 200000 --fs-windows 200000`, BATH_HIP_LANES=1, averaged per kernel "
             "over its dispatches (tools/prof_round2.sh, tools/pmc_summary.py).  <k, 0> = default mode (table log-sums, wavefront scans), <k, 2> = "
             "strict mode (serial order).  hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB as MI355X_MICROARCH.md prescribes for gfx950.",
    "_reading": "End of round 2 (straight-line rows, zero-padded log-sum table, 4 waves per SIMD for the parsers and Backward): the 3-codon parsers run at "
                "30-34 % VALU busy and 21-31 % LDS busy on this 1/5-size block (~1 wave per SIMD), the 5-codon kernels at 3-10 %; HBM traffic is far from "
                "the bound everywhere.  They remain bound by the latency of a wave's dependent chain per row times the rows of the longest window, "
                "and -- once a launch has more windows than the 4096 wave slots 64 KB of LDS table per block allow -- by those slots.",
    "kernels": fs}, open(dst + "/%s_fs_pmc.json" % rnd, "w"), indent=1, sort_keys=True)
name = next(k for k in kernels if "ssv_orf_kernel" in k)
c = kernels[name]
c = c.get("counters", c)
hbm = c["hbm_bytes_per_launch"]
json.dump({"kernel": name.split("(")[0], "workload": "200000 x 1000 nt windows (1/5 of the bench block), BATH_HIP_LANES=1",
           "note": "counters scale linearly with the number of windows; x5 = one launch over the 10^6-window bench block",
           "hbm_bytes_per_launch_200k": hbm, "hbm_bytes_per_launch_full_block": hbm * 5.0, "counters": c},
          open(dst + "/%s_ssv_orf_pmc.json" % rnd, "w"), indent=1, sort_keys=True)
print("wrote", dst + "/%s_fs_pmc.json" % rnd, ";", dst + "/%s_ssv_orf_pmc.json" % rnd, hbm)
