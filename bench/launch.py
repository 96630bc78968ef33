"""`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks ourselves."""
from __future__ import annotations

import os
import socket
import subprocess
import sys

import torch


def self_launch(args, script):
    """One process per GPU under torch.distributed.run on 127.0.0.1 (the reference fans a batch out over its DP workers the same way:
    /root/reference/roll/distributed/scheduler/decorator.py:106-181); returns the children's exit code, or None when this process is a rank
    (or N == 1) and runs the bench itself.  Refuses (exit code 2) when the node has fewer GPUs than ranks, unless SR_DIST_BACKEND=gloo asks
    for the host-staged development layout in which ranks share devices."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and os.environ.get("SR_DIST_BACKEND") != "gloo":
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs (one rank per GPU over RCCL), this node has {have}; "
              f"set SR_DIST_BACKEND=gloo to let ranks share devices on a development box", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # N schedulers + N poll loops share this host: give every rank its share of the cores for its intra-op pool (torch.distributed.run would
    # set OMP_NUM_THREADS=1; the collator and the PNG writers of the pipeline use a few threads), never more than 8
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or args.gpus) // args.gpus))))
    return subprocess.call(cmd, env=env)
