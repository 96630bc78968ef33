"""Data-parallel sharding of tiles over the GPUs of one node.

The reference shards a batch over its ``actor_infer`` workers in contiguous ``np.array_split``-sized chunks and
concatenates results in rank order (/root/reference/roll/distributed/scheduler/decorator.py:106-181,
protocol.py:550-617).  Here: one process per GPU (torchrun), no data-path collective while generating, and ONE
all-gather of the (small) results at the end over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in CPU tests)."""
from __future__ import annotations

import os
from typing import List

import torch
import torch.distributed as dist


def split_sizes(n: int, world: int) -> List[int]:
    """Sizes of np.array_split(range(n), world): the first n % world chunks get one extra element."""
    q, r = divmod(n, world)
    return [q + 1 if i < r else q for i in range(world)]


def shard_range(n: int, rank: int, world: int):
    sizes = split_sizes(n, world)
    start = sum(sizes[:rank])
    return start, start + sizes[rank]


def init_distributed(backend: str | None = None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("SR_DIST_BACKEND") or \
                ("nccl" if torch.cuda.is_available() and torch.cuda.device_count() >= world else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():      # more ranks than GPUs (development box): share the devices, exchange over gloo
            local = local % torch.cuda.device_count()
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Concatenates the per-rank row blocks (array_split sizes) in rank order; every rank gets the full tensor."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = split_sizes(n_total, world)
    mx = max(sizes)
    dev = local.device
    xdev = torch.device("cpu") if dist.get_backend(group) == "gloo" else dev     # gloo exchanges host tensors
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=xdev)
    pad[: local.shape[0]] = local.to(xdev)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0).to(dev)


def decode_with_logits_gather(step_fn, first_logits: torch.Tensor, n_new: int, n_total: int, group=None):
    """BASELINE.json north_star's literal exchange ("RCCL all-gather of logits over xGMI only"), kept as a verification
    mode: every decode step all-gathers the float32 last-position logits [B_local, V] of all ranks, every rank takes the
    greedy token of ALL tiles from the gathered rows and feeds its own rows back through ``step_fn(ids) -> (logits,
    engine_greedy_ids)`` (Engine.decode_step).  Returns (tokens int64 [n_total, n_new], number of positions where the
    engine's own on-device argmax disagreed with the argmax of the gathered logits -- must be 0).
    Costs 608 KB per tile per step on the wire; the default path gathers 1 KB of results per tile once."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo = sum(split_sizes(n_total, world)[:rank])
    b_local = first_logits.shape[0]
    toks, mismatches = [], 0
    logits, own = first_logits, None
    for i in range(n_new):
        allv = all_gather_rows(logits, n_total, group)                # [n_total, V]
        pick = allv.argmax(dim=-1)
        if own is not None:
            mismatches += int((pick[lo:lo + b_local] != own).sum())
        toks.append(pick)
        if i + 1 < n_new:
            logits, own = step_fn(pick[lo:lo + b_local].contiguous())
    return torch.stack(toks, dim=1), mismatches


def all_reduce_max(value: float, device=None) -> float:
    """max over ranks of a host scalar (the bench's timing rule)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    dev = torch.device("cpu") if dist.get_backend() == "gloo" or device is None else device
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()
