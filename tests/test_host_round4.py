"""CPU tests of the round-4 host logic: ranks with different numbers of batches in request-level mode, bench.py's
own N-rank launch / refusal, abort propagation of the cross-rank dispatcher."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_UNEVEN_WORKER = r"""
import os, sys, time, queue
import numpy as np, torch
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
from socioreasoner_amd import dp, hostops
from roll.distributed.scheduler.generate_scheduler import GenerateScheduler
from roll.distributed.scheduler.protocol import DataProto
from roll.distributed.strategy.mi355x_strategy import Mi355xStrategy
from roll.pipeline.base_worker import ActorWorker
from socioreasoner_amd.config import geometry_tiny
from socioreasoner_amd.textproc import ByteTokenizer
from test_host_cpu import _tiny_cfg

rank, world, _ = dp.init_distributed("gloo")
tok = ByteTokenizer(geometry_tiny())
cfg = _tiny_cfg({tmp!r}, prompt_length=12, response_length=12)
cfg["generate_opt_level"] = 1
cfg["rpc_timeout"] = 60


class Fake(Mi355xStrategy):
    max_batch = 1

    def initialize(self, model_provider=None):
        self.command_queue, self.tokenizer = queue.Queue(), tok
        self.served = 0

    def generate(self, batch, generation_config):
        ids, mask = batch.batch["input_ids"], batch.batch["attention_mask"]
        rows = []
        for r, m in zip(ids, mask):
            p = r[m.bool()].tolist()
            rows.append([(7 * t + 3) % 250 for t in p[:3]] + [tok.eos_token_id])
        self.served += len(rows)
        time.sleep(0.005 * len(rows))
        out = hostops.gather_outputs_to_pad_tensor(rows, generation_config["pad_token_id"], device=ids.device)
        return hostops.concatenate_input_and_output(ids, out, 1)


w = ActorWorker(cfg.actor_infer, cfg, rank, world, 0, "actor_infer")
w.strategy = Fake(w)
w.strategy.initialize()
w.tokenizer = tok

# the pipeline's layout: 5 samples over 2 ranks (np.array_split sizes 3 / 2) walked in batches of 2 -> rank 0 owns TWO batches, rank 1 ONE
n_samples, bs = 5, 2
lo, hi = dp.shard_range(n_samples, rank, world)
rng = np.random.default_rng(7)
allp = rng.integers(1, 250, (n_samples, 6))


def batch_of(idx):
    ids = torch.full((len(idx), 12), tok.pad_token_id, dtype=torch.long)
    mask = torch.zeros(len(idx), 12, dtype=torch.long)
    for j, i in enumerate(idx):
        ids[j, 6:] = torch.from_numpy(allp[i])
        mask[j, 6:] = 1
    pos = (mask.cumsum(-1) - 1).clamp(min=0)[:, None, :].repeat(1, 3, 1)
    return DataProto(batch={{"input_ids": ids, "attention_mask": mask, "position_ids": pos}}, non_tensor_batch={{}})


sched = GenerateScheduler()
mine = list(range(lo, hi))
batches = [mine[i:i + bs] for i in range(0, len(mine), bs)]
most = max(-(-s // bs) for s in dp.split_sizes(n_samples, world))
assert (len(batches), most) == ((2, 2) if rank == 0 else (1, 2))
got = {{}}
for b in batches:
    out = sched.generate(batch_of(b), w, cfg)
    for j, i in enumerate(b):
        got[i] = out.batch["responses"][j, :4].tolist()
for _ in range(most - len(batches)):            # what SocioSegInferPipeline.run does after its own last batch
    sched.join_idle_round(w, cfg)
for i in mine:
    assert got[i] == [(7 * int(t) + 3) % 250 for t in allp[i][:3]] + [tok.eos_token_id], (rank, i, got[i])
dp.barrier()
print("ok", rank, "served", w.strategy.served)
"""


def test_request_level_rounds_with_uneven_batch_counts_gloo(tmp_path):
    """at generate_opt_level 1 every dispatch round is collective.  5 samples on 2 ranks at batch 2 give rank 0 two
    batches and rank 1 one; without ``join_idle_round`` rank 0's second round waits for a peer that never comes (TimeoutError after
    rpc_timeout).  Here the short rank joins with no requests of its own -- and serves part of rank 0's last batch."""
    script = tmp_path / "w.py"
    script.write_text(_UNEVEN_WORKER.format(root=ROOT, tmp=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") == 2


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus 2` outside torchrun must never run one rank and print an N = 1 line under an N = 2 flag (VERDICT round 3,
    "multi-GPU launch hazard"): with fewer than 2 GPUs (this container has none) and no SR_DIST_BACKEND=gloo it exits 2 with a message."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SR_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return      # (a multi-GPU node: nothing to refuse)
    assert r.returncode == 2 and "needs 2 visible GPUs" in r.stderr and not r.stdout.strip(), (r.returncode, r.stdout[-300:], r.stderr[-300:])


def test_dispatch_abort_reaches_waiting_ranks():
    """a failure on one rank sets 'abort' in the round's store prefix; another rank blocked in a store wait raises
    within a fraction of a second instead of polling until the round's timeout.  Consumed keys are deleted."""
    import threading
    import time
    from datetime import timedelta
    import pytest
    from torch.distributed import TCPStore
    from socioreasoner_amd.dispatch import CrossRankDispatcher
    store = TCPStore("127.0.0.1", 29549, 1, True, timeout=timedelta(seconds=30))
    a = CrossRankDispatcher(store, 0, 2, round_id=1, timeout_s=60.0)
    b = CrossRankDispatcher(store, 1, 2, round_id=1, timeout_s=60.0)
    threading.Timer(0.3, a._abort).start()
    t0 = time.monotonic()
    with pytest.raises(RuntimeError, match="aborted"):
        b._wait("never/written")
    assert time.monotonic() - t0 < 5.0
    a._set("payload/3", b"x")
    assert b._wait("payload/3") == b"x"
    b._drop("payload/3")
    assert not store.check(["sr_dispatch/1/payload/3"])
