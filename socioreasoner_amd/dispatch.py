"""Request-level dispatch ACROSS the torchrun ranks (SURVEY.md row A15 / section 8(E) "Partitioning").

The reference's ``GenerateScheduler`` is one Ray actor that owns the whole batch: at ``generate_opt_level: 1`` it hands single
requests to the DP worker with the fewest requests in flight (ties -> lowest rank), never more than ``max_running_requests``
(128) per worker, and collects the answers through ``report_response``; the output is re-sorted by prompt id
(/root/reference/roll/distributed/scheduler/generate_scheduler.py:57, 180-187, 195-299, 301-334).

Here the workers are torchrun ranks and every rank holds its own contiguous shard of the batch (socioreasoner_amd.dp), so
the same policy is run over the process group's key-value store (c10d TCPStore -- control traffic only; no collective):

  * rank 0 runs the dispatcher: requests are taken in GLOBAL prompt order, each goes to the rank with the fewest requests in
    flight (sent - reported done; ties -> lowest rank) that is below the cap; ``assign/<gid>`` names the rank and the rank's
    mailbox ``mbox/<rank>/<k>`` receives the request id;
  * a request assigned to another rank than its owner travels as a pickled single-request ``DataProto`` (``payload/<gid>``, device
    tensors moved to the host first) -- what the reference moves through Ray's object store;
  * every rank feeds its mailbox into its OWN engine's request loop (``ActorWorker.add_request(ADD)``), reports every finished
    request (``done/<rank>`` counter; the token ids go back to the owner as ``result/<gid>`` when the owner is another rank);
  * every rank returns the answers of ITS OWN requests in its local order, whoever generated them.

With the reference's cap of 128 a batch of <= 128 x world requests is dealt out at once, interleaved over the ranks; a cap at the
engines' row count makes it work-conserving for skewed answer lengths (tests/test_host_cpu.py shows both).
"""
from __future__ import annotations

import pickle
import threading
import time
from typing import Callable, Dict, List, Optional

import torch


def default_store():
    """The store behind the default process group (TCPStore under torchrun)."""
    import torch.distributed as dist
    from torch.distributed import distributed_c10d as c10d
    if not dist.is_initialized():
        raise RuntimeError("cross-rank dispatch needs an initialised torch.distributed process group")
    return c10d._get_default_store()


def _to_host(obj):
    """device tensors -> host tensors, recursively through dict / list / tuple / object ndarray (a request's images may be
    device-resident: the stage-2 images are rendered on the GPU)."""
    import numpy as np
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_host(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_host(v) for v in obj)
    if isinstance(obj, np.ndarray) and obj.dtype == object:
        out = np.empty(obj.shape, dtype=object)
        for i, v in enumerate(obj.flat):
            out.flat[i] = _to_host(v)
        return out
    return obj


class CrossRankDispatcher:
    """One generate call's worth of request-level dispatch.  Every rank of the group constructs it with the same ``round_id`` and
    calls ``run`` with its own requests; ``run`` returns {local index: token ids}."""

    def __init__(self, store, rank: int, world: int, round_id: int, max_running_requests: int = 128, poll_s: float = 0.002,
                 timeout_s: float = 3600.0):
        from torch.distributed import PrefixStore
        self.store = PrefixStore(f"sr_dispatch/{int(round_id)}/", store)
        self.rank, self.world, self.cap, self.poll_s, self.timeout_s = int(rank), int(world), int(max_running_requests), poll_s, timeout_s
        self.errors: List[BaseException] = []
        self.stats = {"sent_to": [0] * self.world, "served_here": 0, "served_for_others": 0, "shipped_out": 0}

    # ------------------------------------------------------------------ store helpers (values are bytes)
    def _set(self, key: str, val) -> None:
        self.store.set(key, val if isinstance(val, (bytes, bytearray)) else str(val))

    def _wait(self, key: str) -> bytes:
        """value of `key` once it exists.  POLLED (check + sleep), never a blocking get: the threads of a process share one store
        connection, and a get that waits for a key inside the client would keep the dispatcher thread of the same process from
        ever writing it."""
        deadline = time.monotonic() + self.timeout_s
        polls = 0
        while not self.store.check([key]):
            polls += 1
            if self.errors or time.monotonic() > deadline:
                raise TimeoutError(f"rank {self.rank}: key {key!r} never appeared")
            if polls % 64 == 0 and self.aborted():
                raise RuntimeError(f"rank {self.rank}: another rank aborted the dispatch round while waiting for {key!r}")
            time.sleep(self.poll_s)
        return self.store.get(key)

    def aborted(self) -> bool:
        """any rank's failure path sets 'abort': the others stop waiting instead of running into the round's timeout"""
        return bool(self.store.check(["abort"]))

    def _abort(self) -> None:
        try:
            self._set("abort", 1)
        except Exception:  # noqa: BLE001
            pass

    def _drop(self, key: str) -> None:
        """a consumed key leaves the store (rank 0's TCPStore would otherwise keep every pickled request of the run)"""
        try:
            self.store.delete_key(key)
        except Exception:  # noqa: BLE001  (stores without delete support: the key simply stays)
            pass

    def _get_int(self, key: str) -> int:
        return int(self._wait(key).decode())

    def _counter(self, key: str) -> int:
        return int(self.store.add(key, 0))

    # ------------------------------------------------------------------ rank 0: the dispatcher (reference :180-187, 195-260)
    def _dispatch(self, sizes: List[int]):
        try:
            owners = [r for r, n in enumerate(sizes) for _ in range(n)]
            sent = [0] * self.world
            deadline = time.monotonic() + self.timeout_s
            for gid in range(len(owners)):
                while True:
                    load = [sent[r] - self._counter(f"done/{r}") for r in range(self.world)]
                    r = min(range(self.world), key=lambda k: (load[k], k))
                    if load[r] < self.cap:
                        break
                    if self.aborted():                    # another rank failed: stop dealing at once instead of spinning into the timeout
                        raise RuntimeError("dispatcher: another rank aborted the round")
                    if time.monotonic() > deadline:
                        raise TimeoutError("dispatcher: no worker below its request cap before the timeout")
                    time.sleep(self.poll_s)
                self._set(f"assign/{gid}", r)
                self._set(f"mbox/{r}/{sent[r]}", gid)
                sent[r] += 1
            for r in range(self.world):
                self._set(f"mbox/{r}/{sent[r]}", -1)          # STOP after the last request of every mailbox
            self.stats["sent_to"] = sent
        except BaseException as e:  # noqa: BLE001
            self.errors.append(e)
            self._abort()                                     # every rank's polling loops check it

    # ------------------------------------------------------------------ every rank
    def run(self, requests: List, add_request: Callable, make_result_sink: Callable[[Callable], None], make_request: Callable,
            alive_check: Optional[Callable] = None, request_id_key: str = "request_id") -> Dict[int, List[int]]:
        """requests: this rank's single-request DataProtos in local order.  add_request(req): hand a request (meta_info[request_id_key]
        already set to the global id) to the local engine's request loop.  make_result_sink(cb): install cb(gid, token_ids) as the
        completion callback of that loop.  make_request(batch, non_tensor_batch, meta_info): rebuild a request that arrived from
        another rank.  Returns {local index: token ids} for all of this rank's own requests."""
        n_local = len(requests)
        self._set(f"n/{self.rank}", n_local)
        sizes = [self._get_int(f"n/{r}") for r in range(self.world)]
        off = sum(sizes[: self.rank])
        owner_of = [r for r, n in enumerate(sizes) for _ in range(n)]
        results: Dict[int, List[int]] = {}
        lock = threading.Lock()
        own_done = threading.Event()
        if n_local == 0:
            own_done.set()

        def on_complete(gid: int, toks: List[int]):
            gid = int(gid)
            if owner_of[gid] == self.rank:
                with lock:
                    results[gid - off] = list(toks)
                    if len(results) == n_local:
                        own_done.set()
            else:
                self._set(f"result/{gid}", pickle.dumps(list(toks)))
                self.stats["served_for_others"] += 1
            self.store.add(f"done/{self.rank}", 1)
            self.stats["served_here"] += 1

        make_result_sink(on_complete)
        threads = []
        if self.rank == 0:
            threads.append(threading.Thread(target=self._dispatch, args=(sizes,), daemon=True))

        def publish():           # owner duty: ship the requests that were assigned elsewhere
            try:
                for i, req in enumerate(requests):
                    tgt = self._get_int(f"assign/{off + i}")
                    if tgt != self.rank:
                        req.meta_info[request_id_key] = off + i
                        meta = {k: v for k, v in req.meta_info.items() if not callable(v)}      # callbacks do not travel
                        blob = pickle.dumps({"batch": _to_host(req.batch), "non_tensor_batch": _to_host(req.non_tensor_batch), "meta_info": meta})
                        self._set(f"payload/{off + i}", blob)
                        self.stats["shipped_out"] += 1
            except BaseException as e:  # noqa: BLE001
                self.errors.append(e)
                self._abort()

        def mailbox():           # worker duty: feed the local engine
            try:
                k = 0
                while True:
                    gid = self._get_int(f"mbox/{self.rank}/{k}")
                    if gid < 0:
                        return
                    if owner_of[gid] == self.rank:
                        req = requests[gid - off]
                    else:
                        d = pickle.loads(self._wait(f"payload/{gid}"))
                        self._drop(f"payload/{gid}")
                        req = make_request(d["batch"], d["non_tensor_batch"], d["meta_info"])
                    req.meta_info[request_id_key] = gid
                    add_request(req)
                    k += 1
            except BaseException as e:  # noqa: BLE001
                self.errors.append(e)
                self._abort()

        threads += [threading.Thread(target=publish, daemon=True), threading.Thread(target=mailbox, daemon=True)]
        for t in threads:
            t.start()
        # own results: generated here (callback) or on another rank (store)
        deadline = time.monotonic() + self.timeout_s
        remote_pending = None
        ticks = 0
        while True:
            ticks += 1
            if self.errors:
                self._abort()
                raise RuntimeError("cross-rank dispatch failed") from self.errors[0]
            if ticks % 8 == 0 and self.aborted():
                raise RuntimeError(f"rank {self.rank}: another rank aborted the dispatch round")
            if alive_check is not None:
                try:
                    alive_check()
                except BaseException:
                    self._abort()
                    raise
            if remote_pending is None and (n_local == 0 or self.store.check([f"assign/{off + i}" for i in range(n_local)])):
                # which of my requests run elsewhere is known once they are all assigned
                remote_pending = [i for i in range(n_local) if self._get_int(f"assign/{off + i}") != self.rank]
            if remote_pending:
                still = []
                for i in remote_pending:
                    if self.store.check([f"result/{off + i}"]):
                        toks = pickle.loads(self.store.get(f"result/{off + i}"))
                        self._drop(f"result/{off + i}")
                        with lock:
                            results[i] = toks
                            if len(results) == n_local:
                                own_done.set()
                    else:
                        still.append(i)
                remote_pending = still
            if own_done.wait(timeout=0.02 if remote_pending else 0.1):
                break
            if time.monotonic() > deadline:
                raise TimeoutError(f"rank {self.rank}: {n_local - len(results)} of {n_local} requests did not complete")
        # the local engine may still be serving other ranks' requests: wait until everything dealt out has been reported
        total = sum(sizes)
        while sum(self._counter(f"done/{r}") for r in range(self.world)) < total:
            if self.errors:
                self._abort()
                raise RuntimeError("cross-rank dispatch failed") from self.errors[0]
            if self.aborted():
                raise RuntimeError(f"rank {self.rank}: another rank aborted the dispatch round")
            if alive_check is not None:
                try:
                    alive_check()
                except BaseException:
                    self._abort()
                    raise
            if time.monotonic() > deadline:
                raise TimeoutError("cross-rank dispatch: not every request was reported done")
            time.sleep(self.poll_s)
        for t in threads:
            t.join(timeout=5)
        return results
