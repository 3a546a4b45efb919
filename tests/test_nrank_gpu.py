"""GPU tier: the N-rank legs of bench.py with REAL kernels on ONE GPU.

The merge the reference does after its workers finish (bathsearch.c:868-921: p7_tophits_Merge, p7_pipeline_Merge, E-values with
the whole search's residue count, p7_tophits_RemoveDuplicates) is what `bench.py --gpus N` reproduces over ranks -- hits
serialized, window -> target shift, per-query exchange to the owner rank, counters all-reduced, nres_before per item.  RCCL refuses
two ranks on one device, so the ranks here share GPU 0 (BATH_BENCH_SHARE_DEVICE=1) and the collectives run on CPU tensors over
gloo (BATH_BENCH_BACKEND=gloo): every rank computes with the HIP kernels, and rank 0 then runs the same search ALONE and compares
-- tables byte for byte (configs[3]), domains record for record incl. score bits and CIGAR strings (configs[2], configs[4]),
counters (all legs).  bench.py is started as a fresh child process: its launcher starts torch.distributed.run before any GPU call.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(n, scaling, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BATH_BENCH_BACKEND="gloo", BATH_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--scaling", scaling,
           "--windows", "20000", "--fs-windows", "20000", "--c4-total-mb", "12", "--c5-total-mb", "30", "--no-cpu-baseline"] + list(extra)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1 and len(lines[0]) < 2000
    full = [l for l in p.stderr.splitlines() if l.startswith("{") and '"residues_per_step"' in l]
    assert len(full) == 1
    return json.loads(lines[0]), json.loads(full[0])


def test_weak_scaling_headline_sums_the_ranks():
    line, out = run_ranks(2, "weak", ["--no-c45", "--no-fs"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["residues_per_step"] == 2 * 1000 * 20000 * 2                      # a block per rank, counters summed
    assert out["hits_gathered"] == out["survivors"]["n_past_msv"] > 0            # every rank's records reached rank 0


@pytest.mark.parametrize("n", [2, 3])
def test_n_rank_search_equals_the_single_rank_search(n):
    scaling = "strong"
    line, out = run_ranks(n, scaling)
    assert out["n_gpus"] == n and out["scaling"] == scaling and out["value"] > 0 and not out.get("plumbing_only")
    # the headline block: ONE block of 20 000 windows over the ranks -- counters and ORF records of the merged search equal the
    # block searched on one rank
    chk = out["strong_scaling_check"]
    assert chk["counters_equal_to_single_rank"] and chk["records_equal_to_single_rank"], chk
    assert out["residues_per_step"] == 2 * 1000 * 20000
    assert line["strong_check"] == {"counters_equal_to_single_rank": True, "records_equal_to_single_rank": True}
    assert out["hits_gathered"] == out["survivors"]["n_past_msv"] > 0

    # configs[3]: 12 queries x 12 Mb as (query, window group) items over the ranks, every query finished on its owner rank:
    # the tables rank 0 prints are the single-rank search's, byte for byte
    c4 = out["c4"]
    assert c4["n_gpus"] == n and sum(c4["items_per_rank"]) == c4["items"] >= 12 and min(c4["items_per_rank"]) >= 1
    assert c4["tables_equal_to_single_rank_search"] is True, (c4["queries_whose_tables_differ"], c4["hits_per_query"], c4["single_rank_hits_per_query"])
    assert c4["hits_per_query"] == c4["single_rank_hits_per_query"] and c4["hits"] > 0
    assert line["c4"]["tables_equal"] is True

    # configs[4]: the 1024-node model with --fs over window shards; configs[2]: the --fs block over window shards
    for leg in ("c5", "fs"):
        r = out[leg]
        assert r["n_gpus"] == n and r["windows_of_gathered_domains_are_global"]
        assert r["domains_equal_to_single_rank_search"] is True, r["single_rank_check"]
        assert r["counters_equal_to_single_rank_search"] is True, r["single_rank_check"]
        assert r["domains_gathered"] == r["single_rank_check"]["single_rank_search"] > 0
        assert line[leg]["domains_equal"] is True and line[leg]["counters_equal"] is True
