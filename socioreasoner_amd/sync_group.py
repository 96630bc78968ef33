"""Named process groups for the trainer -> engine weight sync (SURVEY.md "next" row N4).

The reference creates one extra process group per (trainer rank -> engine ranks) broadcast tree, OUTSIDE the default group:
the members rendezvous on ``tcp://master_addr:master_port`` (rank 0 = the trainer rank that owns the weights, ranks 1.. = the
engine ranks in the order of ``comm_plan_args["tgt_devices"]``) and the group's store keys live under the group's name
(/root/reference/roll/utils/collective/collective.py:13-75, pg_utils.py:10-75).  An engine rank that wants to be a drop-in
member of such a group has to join with exactly that recipe -- same rendezvous URL, same key prefix, same backend
constructor -- or the trainer's half of the handshake never meets ours.  ``torch.distributed`` has no public call for "a
second, unrelated world", so this module uses the same c10d building blocks ``init_process_group`` itself is made of.
"""
from __future__ import annotations

import datetime
from typing import Dict

import torch
import torch.distributed as dist

_groups: Dict[str, "dist.ProcessGroup"] = {}


def join_named_group(name: str, backend: str, world_size: int, rank: int, master_addr: str, master_port: int,
                     timeout_s: float = 1800.0):
    """Join (or, as rank 0, host) the group ``name``; returns a ProcessGroup usable with dist.broadcast / all_reduce."""
    if not name:
        raise ValueError("a sync group needs a name")
    if name in _groups:
        raise RuntimeError(f"sync group {name!r} was already joined by this process")
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    from torch.distributed import distributed_c10d as c10d
    timeout = datetime.timedelta(seconds=timeout_s)
    store, rank, world_size = next(iter(c10d.rendezvous(f"tcp://{master_addr}:{int(master_port)}", rank, world_size, timeout=timeout)))
    store.set_timeout(timeout)
    scoped = c10d.PrefixStore(name, store)                       # keys of different groups on one store never collide
    opt_kw = "backend_options" if tuple(int(x) for x in torch.__version__.split("+")[0].split(".")[:2]) >= (2, 6) else "pg_options"
    pg, _ = c10d._new_process_group_helper(world_size, rank, [], c10d.Backend(backend), scoped, group_name=name, timeout=timeout,
                                           **{opt_kw: None})
    c10d._world.pg_group_ranks[pg] = {r: r for r in range(world_size)}     # group rank == global rank inside this little world
    _groups[name] = pg
    return pg


def get_group(name: str):
    return _groups.get(name)


def leave_named_group(name: str):
    pg = _groups.pop(name, None)
    if pg is not None:
        try:
            dist.destroy_process_group(pg)
        except Exception:  # noqa: BLE001
            pass
