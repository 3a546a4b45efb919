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

STAT_FIELDS = [n for n, _ in PipelineStats._fields_]


def shard_range(n_items, rank, world):
    """Contiguous block partition of target blocks, like the reference's block queue dealt in order."""
    per, rem = divmod(n_items, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


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


def gather_results(res, window_offset, dst=0, device="cpu"):
    """p7_tophits_Merge: variable-length gather of the per-rank ORF records on rank <dst>; window indices
    are made global by adding each rank's shard offset."""
    res = res.copy()
    res["window"] += window_offset
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return res
    world, rank = dist.get_world_size(), dist.get_rank()
    raw = torch.from_numpy(res.view(np.uint8).reshape(-1).copy()).to(device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([raw.numel()], dtype=torch.int64, device=device))
    mx = int(max(int(c.item()) for c in counts))
    pad = torch.zeros(mx, dtype=torch.uint8, device=device)
    pad[: raw.numel()] = raw
    bufs = [torch.empty(mx, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(bufs, pad)
    if rank != dst:
        return None
    parts = [np.frombuffer(bufs[r][: int(counts[r].item())].cpu().numpy().tobytes(), dtype=ORF_RESULT_DTYPE) for r in range(world)]
    return np.concatenate(parts)


def gather_bytes(data, dst=0, device="cpu"):
    """Variable-length gather of one bytes object per rank; the list on rank <dst>, None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [data]
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(data)], dtype=torch.int64, device=device))
    mx = max(1, int(max(int(c.item()) for c in counts)))
    pad = torch.zeros(mx, dtype=torch.uint8, device=device)
    if len(data):
        pad[: len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    bufs = [torch.empty(mx, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(bufs, pad)
    if rank != dst:
        return None
    return [bytes(bufs[r][: int(counts[r].item())].cpu().numpy().tobytes()) for r in range(world)]


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
