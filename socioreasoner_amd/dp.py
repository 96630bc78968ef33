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


# ranks 1..N-1 of the bench wait in a barrier while rank 0 alone times the static reference, the batch-1 run and the CPU baselines
# (a minute or two): the collective timeout must not be what ends that wait
_PG_TIMEOUT = __import__("datetime").timedelta(minutes=30)


def init_distributed(backend: str | None = None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).  Returns (rank, world, local_rank).

    One process per GPU and RCCL (torch backend "nccl") is THE multi-GPU path.  gloo is never chosen silently on a GPU
    box: it has to be asked for (argument or SR_DIST_BACKEND=gloo -- CPU tests, or a development box with fewer GPUs than
    ranks, where the ranks share devices and exchange through host memory).  SR_FORCE_DIST=1 opens the process group even
    for a single rank, so that the RCCL code path can be exercised on a one-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    force = os.environ.get("SR_FORCE_DIST", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        backend = backend or os.environ.get("SR_DIST_BACKEND")
        have_gpu = torch.cuda.is_available()
        if backend is None:
            if not have_gpu:
                backend = "gloo"
            elif torch.cuda.device_count() >= world:
                backend = "nccl"
            else:
                raise RuntimeError(f"{world} ranks but {torch.cuda.device_count()} visible GPU(s): one process per GPU over RCCL is the "
                                   f"supported layout; set SR_DIST_BACKEND=gloo to share devices on a development box")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            if torch.cuda.device_count() < world:
                raise RuntimeError(f"backend nccl (RCCL) needs one GPU per rank: {world} ranks, {torch.cuda.device_count()} GPU(s)")
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"), timeout=_PG_TIMEOUT)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=_PG_TIMEOUT)
    # more ranks than GPUs (the development layout over gloo): the ranks share the devices.  Also when the process group was set up by an earlier call
    # (roll.distributed.scheduler.initialize.init before the pipeline's own): the device index must not depend on who initialised first
    if torch.cuda.is_available() and not (dist.is_initialized() and dist.get_backend() == "nccl"):
        local = local % max(torch.cuda.device_count(), 1)
    return rank, world, local


_gather_bufs: dict = {}


def _gather_equal(pad: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """all-gather of equally sized blocks -> [world, *pad.shape].  RCCL: ONE all_gather_into_tensor into a pre-allocated,
    reused device buffer (no list of views, no per-call allocation, no host staging); gloo: host tensors."""
    if dist.get_backend(group) == "gloo":
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        return torch.stack(bufs, dim=0)
    key = (tuple(pad.shape), pad.dtype, pad.device, world, id(group))
    out = _gather_bufs.get(key)
    if out is None:
        out = _gather_bufs[key] = torch.empty((world,) + tuple(pad.shape), dtype=pad.dtype, device=pad.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    return out


def exchange_info(group=None) -> dict:
    """What the data-path exchange actually ran on -- goes into the bench line so that a scaling record can be checked
    against "RCCL saw N ranks" (and, with NCCL_DEBUG=INFO set by the launcher, which transports its channels use)."""
    if not dist.is_initialized():
        return {"backend": "none (single process)", "nranks": 1}
    info = {"backend": dist.get_backend(group), "nranks": dist.get_world_size(group)}
    if info["backend"] == "nccl":
        try:
            v = torch.cuda.nccl.version()
            info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
        except Exception:  # noqa: BLE001
            pass
        info["collective"] = "all_gather_into_tensor (pre-allocated device buffer)"
        path = os.environ.get("SR_RCCL_LOG")
        if path and os.path.exists(path):
            txt = open(path, errors="replace").read()
            info["log_nranks"] = sorted({int(t.split()[0]) for t in txt.split("nranks ")[1:] if t.split() and t.split()[0].isdigit()})
            info["channels_via_p2p"] = txt.count(" via P2P/")
            info["channels_via_shm"] = txt.count(" via SHM/")
            info["channels_via_net"] = txt.count(" via NET/")
    return info


def all_gather_rows(local: torch.Tensor, n_total: int, group=None, sizes=None) -> torch.Tensor:
    """Concatenates the per-rank row blocks in rank order; every rank gets the full tensor.  Per-rank row counts are the
    array_split sizes of ``n_total`` unless ``sizes`` gives them explicitly (e.g. n_ret rows per sample: the SAMPLES are
    array_split over ranks, so rank r holds split_sizes(n_samples)[r] * n_ret rows, not split_sizes(n_samples * n_ret)[r])."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = split_sizes(n_total, world) if sizes is None else [int(v) for v in sizes]
    if len(sizes) != world or sum(sizes) != n_total or local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError(f"all_gather_rows: rank {dist.get_rank(group)} holds {local.shape[0]} rows, per-rank sizes {sizes}, total {n_total}")
    mx = max(sizes)
    dev = local.device
    xdev = torch.device("cpu") if dist.get_backend(group) == "gloo" else dev     # gloo exchanges host tensors
    if all(sz == mx for sz in sizes):
        # (clone: on the RCCL path _gather_equal hands out its process-wide exchange buffer, which the next call overwrites)
        return _gather_equal(local.to(xdev).contiguous(), world, group).reshape((n_total,) + tuple(local.shape[1:])).to(dev, copy=True)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=xdev)
    pad[: local.shape[0]] = local.to(xdev)
    full = _gather_equal(pad, world, group)
    return torch.cat([full[r, :sz] for r, sz in enumerate(sizes)], dim=0).to(dev)


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
