"""Multi-GPU layout of the bathsearch hot path: one process per GPU, targets sharded, no data-path collective.

The reference parallelises by handing blocks of target sequence to POSIX worker threads that each own a
cloned profile and a private P7_PIPELINE / P7_TOPHITS, merged at the end (bathsearch.c:814-844, 886-908).
The same shape over RCCL: the query profile is broadcast once from rank 0 (a few tens of KB), every rank
scores its own shard of DNA windows independently, and the per-rank hit lists and pipeline counters are
gathered on rank 0 (p7_tophits_Merge / p7_pipeline_Merge, p7_pipeline.c:736).  torch.distributed is
plumbing only (backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist

import ctypes as C

from . import ORF_RESULT_DTYPE, FsDomain, PipelineStats
from . import lib as _lib

STAT_FIELDS = [n for n, _ in PipelineStats._fields_]


def shard_range(n_items, rank, world):
    """Contiguous block partition of target blocks, like the reference's block queue dealt in order."""
    lo, hi = C.c_int64(0), C.c_int64(0)
    _lib().bath_dist_shard_range(int(n_items), int(rank), int(world), C.byref(lo), C.byref(hi))    # the C surface a C host uses (include/bath_hip.h)
    return lo.value, hi.value


BLOCK_LENGTH = 262144        # BATH_MAX_RESIDUE_COUNT, the default --block_length (bathsearch.c:839)


def split_targets(lengths, max_length, block_length=BLOCK_LENGTH):
    """The windows esl_sqio_ReadWindow hands to the workers (bathsearch.c:1060, 1099): a target longer than <block_length>
    is read in windows of at most block_length new nucleotides, each after the first preceded by a context of
    C = 3 * max_length nucleotides of the previous window, so that no ORF is lost at a boundary.

    Returns a list of (seqidx, start0, n, C): window = target[start0 : start0 + n], its first C nucleotides are context.
    A hit found at window coordinate p lies at target coordinate start0 + p on either strand."""
    C_ = 3 * int(max_length)
    out = []
    for idx, L in enumerate(lengths):
        L = int(L)
        pos = 0
        while True:
            c = 0 if pos == 0 else C_
            n_new = min(block_length, L - pos)
            out.append((idx, pos - c, n_new + c, c))
            pos += n_new
            if pos >= L:
                break
    return out


def broadcast_bytes(data, src=0, device="cpu"):
    """Broadcast a bytes object (the query model file) from <src> to every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return data
    n = torch.tensor([len(data) if dist.get_rank() == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    if dist.get_rank() == src:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src)
    return bytes(buf.cpu().numpy().tobytes())


def reduce_stats(stats, device="cpu"):
    """p7_pipeline_Merge: sum the counters over ranks (returned as a dict on every rank)."""
    vals = torch.tensor([getattr(stats, f) for f in STAT_FIELDS], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
    return dict(zip(STAT_FIELDS, [int(v) for v in vals.cpu()]))


def exchange_bytes(by_dest, device="cpu"):
    """Variable-length exchange of bytes objects: {destination rank: payload} on every rank -> {source rank: payload} on every rank
    (only non-empty payloads travel; a rank's payload to itself does not leave the process).  The byte counts go in ONE all-gather;
    then every send and every receive of the exchange is posted at once (batch_isend_irecv: a grouped launch under RCCL), so the
    streams of N - 1 ranks to one receiver move side by side, each on its own xGMI link, instead of one rank after the other."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return {0: by_dest[0]} if by_dest.get(0) else {}
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = np.zeros(world, dtype=np.int64)
    for d, payload in by_dest.items():
        if not 0 <= int(d) < world:
            raise ValueError("exchange_bytes: no rank %r in a world of %d" % (d, world))
        mine[int(d)] = len(payload)
    counts = [torch.zeros(world, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.from_numpy(mine).to(device))
    counts = torch.stack(counts).cpu().numpy()                      # counts[src][dst]
    ops, bufs, keep = [], {}, []
    for src in range(world):
        k = int(counts[src][rank])
        if src != rank and k:
            bufs[src] = torch.empty(k, dtype=torch.uint8, device=device)
            ops.append(dist.P2POp(dist.irecv, bufs[src], src))
    for d, payload in by_dest.items():
        if int(d) != rank and len(payload):
            t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, int(d)))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    out = {src: bytes(b.cpu().numpy().tobytes()) for src, b in bufs.items()}
    if by_dest.get(rank):
        out[rank] = by_dest[rank]
    return out


def gather_bytes(data, dst=0, device="cpu"):
    """Variable-length gather of one bytes object per rank ON RANK <dst> ONLY (p7_tophits_Merge's direction: workers -> master):
    every rank's byte count in one all-gather, then every payload point to point with all the receives posted at once
    (exchange_bytes); nothing is replicated to the other ranks.  Returns the list of payloads on rank <dst>, None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [data]
    got = exchange_bytes({dst: data}, device)
    if dist.get_rank() != dst:
        return None
    return [got.get(r, b"") for r in range(dist.get_world_size())]


def gather_results(res, window_offset, dst=0, device="cpu"):
    """p7_tophits_Merge: variable-length gather of the per-rank ORF records on rank <dst>; window indices
    are made global by adding each rank's shard offset."""
    res = res.copy()
    res["window"] += window_offset
    parts = gather_bytes(res.view(np.uint8).reshape(-1).tobytes(), dst, device)
    if parts is None:
        return None
    return np.concatenate([np.frombuffer(p, dtype=ORF_RESULT_DTYPE) for p in parts])


def gather_domains(domains, window_offset, dst=0, device="cpu"):
    """p7_tophits_Merge for the hits proper: every rank's bath_fs_domain records with their CIGAR strings arrive on rank
    <dst> (a list of FsDomain with .cigar set and window indices made global); None on the other ranks."""
    sz = C.sizeof(FsDomain)
    blob = bytearray()
    for d in domains:
        x = FsDomain()
        C.memmove(C.byref(x), C.byref(d), sz)
        x.window += window_offset
        cig = d.cigar.encode()
        blob += bytes(x) + len(cig).to_bytes(4, "little") + cig
    parts = gather_bytes(bytes(blob), dst, device)
    if parts is None:
        return None
    out = []
    for part in parts:
        p = 0
        while p < len(part):
            x = FsDomain.from_buffer_copy(part[p : p + sz]); p += sz
            n = int.from_bytes(part[p : p + 4], "little"); p += 4
            x.cigar = part[p : p + n].decode(); p += n
            out.append(x)
    return out


def max_over_ranks(x, device="cpu"):
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---------------------------------------------------------------------------------------------------------------------------
# A multi-query database over N ranks (BASELINE configs[3]).  bathsearch loops over the queries of the database, and for every
# query hands the target's blocks to its workers, merges their hit lists and finishes the query (bathsearch.c:737-844, 868-921).
# One process per GPU: the whole database is broadcast ONCE, the unit dealt to the ranks is a (query, group of consecutive
# windows of the target) pair -- per query the fixed cost of a search (profile conversion, a dozen small launches, domain
# definition) is paid by one or two ranks on a large block instead of by every rank on a small one -- and per query the hits
# and the 14 counters travel to rank 0, which finishes each query exactly as the single-rank search does.
# ---------------------------------------------------------------------------------------------------------------------------

def query_items(n_windows_by_query, world, items_per_rank=2):
    """[(query, lo, hi)]: each query's windows [0, n) cut into G consecutive groups, G the smallest count that gives every rank
    about <items_per_rank> items (G = 1 while there are at least that many queries per rank)."""
    return _items(n_windows_by_query, None, world, items_per_rank)


def item_cost(M, n_nt):
    """Estimated milliseconds of one (query, window group) item of a plain (no --fs) search on an MI355X, for the longest-first deal:
    a fixed part that grows with the model (the latency chains of a query's cascade and domain stage: 1.5 ms + 9 us per node) and a
    throughput part, 7.5e-11 ms per node and nucleotide (the cascade's rate: 2 strands x 0.75 ORF residues per nt x M cells at 2e13
    cells/s).  The throughput term is fixed from the cascade's measured rate, the other two are a least-squares fit to the 13 items
    of the 100 Mb x 12-model job measured alone (profiles/r05_c4_items.txt: 2.75 ms at 56 nodes ... 7.6 ms for HALF the genome at
    459 nodes): within 13 % of every item but one (tests/test_dist_cpu.py pins the ordering)."""
    return 1.5 + 0.009 * M + 7.5e-11 * M * n_nt


def query_items_weighted(n_windows_by_query, costs_by_query, world, items_per_rank=3):
    """[(query, lo, hi)] like query_items, but a query gets groups in proportion to its share of the work: round(share x T) of them,
    at least one, T = max(queries, items_per_rank x world).  With worker contexts running a rank's items side by side the longest
    item is the critical path, so the 459-node model of a 12-model database is cut in two even on one rank while the short models
    stay whole (every extra item repeats a search's fixed cost).  The same on every rank, no communication."""
    return _items(n_windows_by_query, costs_by_query, world, items_per_rank)


def _items(n_windows_by_query, costs_by_query, world, items_per_rank):
    """bath_dist_items (libbathhip: the cut a C host computes)."""
    from . import DistItem
    nq = len(n_windows_by_query)
    nw = (C.c_int64 * max(nq, 1))(*[int(x) for x in n_windows_by_query])
    costs = (C.c_double * max(nq, 1))(*[float(x) for x in costs_by_query]) if costs_by_query is not None else None
    n = _lib().bath_dist_items(nw, costs, nq, int(world), int(items_per_rank), None, 0)
    assert n >= 0
    arr = (DistItem * max(n, 1))()
    _lib().bath_dist_items(nw, costs, nq, int(world), int(items_per_rank), arr, n)
    return [(int(arr[k].query), int(arr[k].lo), int(arr[k].hi)) for k in range(n)]


def deal(costs, world):
    """Owner rank of every item: longest processing time first onto the least loaded rank (ties: lowest rank); the same on every
    rank, no communication."""
    n = len(costs)
    cs = (C.c_double * max(n, 1))(*[float(x) for x in costs])
    owner = (C.c_int32 * max(n, 1))()
    assert _lib().bath_dist_deal(cs, n, int(world), owner) == 0                   # bath_dist_deal (libbathhip)
    return [int(owner[k]) for k in range(n)]


def gather_query_domains(by_query, dst=0, device="cpu"):
    """p7_tophits_Merge per query in ONE variable-length gather: {query: [FsDomain with .cigar]} of every rank -> the union per
    query on rank <dst> (None elsewhere).  Window indices must already be the query's own (global) window numbers."""
    sz = C.sizeof(FsDomain)
    blob = bytearray()
    for q in sorted(by_query):
        for d in by_query[q]:
            cig = d.cigar.encode()
            blob += int(q).to_bytes(4, "little") + bytes(d) + len(cig).to_bytes(4, "little") + cig
    parts = gather_bytes(bytes(blob), dst, device)
    if parts is None:
        return None
    out = {}
    for part in parts:
        p = 0
        while p < len(part):
            q = int.from_bytes(part[p : p + 4], "little"); p += 4
            x = FsDomain.from_buffer_copy(part[p : p + sz]); p += sz
            n = int.from_bytes(part[p : p + 4], "little"); p += 4
            x.cigar = part[p : p + n].decode(); p += n
            out.setdefault(q, []).append(x)
    return out


def gather_query_hits(by_query, dst=0, device="cpu"):
    """gather_query_domains without a Python object per hit: {query: HitArray} of every rank -> {query: HitArray} (the union, CIGAR
    offsets rebased) on rank <dst>, None elsewhere.  A rank's payload is its record arrays and CIGAR pools as they left the library."""
    from . import HitArray
    blob = bytearray()
    for q in sorted(by_query):
        blob += int(q).to_bytes(4, "little") + by_query[q].to_bytes()
    parts = gather_bytes(bytes(blob), dst, device)
    if parts is None:
        return None
    got = {}
    for part in parts:
        p = 0
        while p < len(part):
            q = int.from_bytes(part[p : p + 4], "little")
            h, p = HitArray.from_bytes(part, p + 4)
            got.setdefault(q, []).append(h)
    return {q: HitArray.concat(v) for q, v in got.items()}


def query_owner(query, world):
    """The rank that finishes <query> (its hit list merged, E-values, duplicates, sorting, the table's text): q mod N, so that the
    serial end of a multi-query job is spread over the ranks instead of sitting on rank 0."""
    return int(query) % int(world)


def exchange_query_hits(by_query, device="cpu"):
    """p7_tophits_Merge per query ON THE QUERY'S OWNER: {query: HitArray} of every rank -> {query: HitArray} (the union over the
    ranks, CIGAR offsets rebased) for the queries this rank owns (query_owner).  One exchange_bytes for all queries."""
    from . import HitArray
    world = dist.get_world_size() if dist.is_initialized() else 1
    by_dest = {}
    for q in sorted(by_query):
        by_dest.setdefault(query_owner(q, world), bytearray()).extend(int(q).to_bytes(4, "little") + by_query[q].to_bytes())
    parts = exchange_bytes({d: bytes(b) for d, b in by_dest.items()}, device)
    got = {}
    for src in sorted(parts):                                      # rank order: the merged list does not depend on arrival order
        part, p = parts[src], 0
        while p < len(part):
            q = int.from_bytes(part[p : p + 4], "little")
            h, p = HitArray.from_bytes(part, p + 4)
            got.setdefault(q, []).append(h)
    return {q: HitArray.concat(v) for q, v in got.items()}


def gather_query_tables(tables, dst=0, device="cpu"):
    """The finished queries travel to the rank that prints: {query: (reported hits, table text)} of every owner -> the same dict
    for all queries on rank <dst> (None elsewhere).  What moves is the table's text, not the hits."""
    blob = bytearray()
    for q in sorted(tables):
        n, text = tables[q]
        t = text.encode()
        blob += int(q).to_bytes(4, "little") + int(n).to_bytes(8, "little") + len(t).to_bytes(8, "little") + t
    parts = gather_bytes(bytes(blob), dst, device)
    if parts is None:
        return None
    out = {}
    for part in parts:
        p = 0
        while p < len(part):
            q = int.from_bytes(part[p : p + 4], "little"); n = int.from_bytes(part[p + 4 : p + 12], "little"); k = int.from_bytes(part[p + 12 : p + 20], "little")
            out[q] = (n, part[p + 20 : p + 20 + k].decode()); p += 20 + k
    return out


def reduce_query_stats(stats_by_query, n_queries, device="cpu"):
    """p7_pipeline_Merge per query in one all-reduce: {query: PipelineStats (this rank's sum over its items)} -> [dict] per query."""
    arr = np.zeros((n_queries, len(STAT_FIELDS)), dtype=np.int64)            # filled with numpy: a torch scalar assignment costs 8 us apiece
    for q, st in stats_by_query.items():
        arr[q] += [int(st[f]) if isinstance(st, dict) else int(getattr(st, f)) for f in STAT_FIELDS]
    if dist.is_initialized() and dist.get_world_size() > 1:
        vals = torch.from_numpy(arr).to(device)
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
        arr = vals.cpu().numpy()
    return [dict(zip(STAT_FIELDS, [int(v) for v in arr[q]])) for q in range(n_queries)]


def gather_floats(x, dst=0, device="cpu"):
    """One float per rank on rank <dst> (per-rank busy times of a leg); None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(x)]
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.get_rank() == dst:
        outs = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(dist.get_world_size())]
        dist.gather(t, outs, dst=dst)
        return [float(o.item()) for o in outs]
    dist.gather(t, None, dst=dst)
    return None
