"""``GenerateScheduler`` (reference: roll/distributed/scheduler/generate_scheduler.py:63-334).

The reference fans a batch out to its DP workers through Ray (level 0) or dispatches single requests to the least
loaded worker and collects them through callbacks (level 1).  Here every torchrun rank already holds its shard of the
batch (socioreasoner_amd.dp): level 0 is one ``generate`` call on the local worker; level 1 splits the batch into
single-prompt requests and -- with more than one rank -- deals them out ACROSS the ranks like the reference does
(least requests in flight first, ``max_running_requests`` per worker, reference :57, 180-187), each rank's request loop
serving what it is handed and every rank getting the answers of its own prompts back in prompt order
(socioreasoner_amd/dispatch.py); with one rank the local request loop serves them all."""
from __future__ import annotations

import threading
from typing import Dict, List

import numpy as np
import torch

from roll.distributed.scheduler.protocol import DataProto
from roll.utils.functionals import GenerateRequestType
from socioreasoner_amd import hostops


def expand_num_return_sequences(data: DataProto, n: int) -> DataProto:
    """Repeat every prompt n times, neighbours adjacent (reference :103-140)."""
    if n == 1:
        return data
    batch = {k: v.repeat_interleave(n, dim=0) for k, v in data.batch.items()}
    nt = {k: np.repeat(v, n) for k, v in data.non_tensor_batch.items()}
    return DataProto(batch=batch, non_tensor_batch=nt, meta_info=dict(data.meta_info))


class GenerateScheduler:
    max_running_requests = 128          # per worker (reference :57)

    def __init__(self):
        self.lock = threading.Lock()
        self.results: Dict[int, List[int]] = {}
        self.done = threading.Event()
        self.round = 0                  # generate calls so far: every rank counts them alike, keys of different calls never meet
        self.sink = None                # cross-rank mode: where finished requests are reported
        self.last_dispatch_stats = None

    def report_response(self, data: DataProto):
        if self.sink is not None:       # cross-rank dispatch: the request may belong to another rank
            self.sink(int(data.meta_info["request_id"]), data.meta_info["output_token_ids"][0])
            return
        with self.lock:
            self.results[int(data.meta_info["request_id"])] = data.meta_info["output_token_ids"][0]
            if len(self.results) == self.expected:
                self.done.set()

    def generate(self, data: DataProto, actor_cluster, pipeline_config) -> DataProto:
        worker = actor_cluster
        ga = dict(worker.worker_config.generating_args or {})
        n = int(ga.get("num_return_sequences", 1) or 1)
        expand = bool(pipeline_config.get("is_num_return_sequences_expand"))
        gc = dict(ga)
        gc["num_return_sequences"] = 1 if expand else n
        gc.setdefault("max_new_tokens", int(pipeline_config.response_length))
        data.meta_info["generation_config"] = gc
        data.batch["prompt_id"] = torch.arange(len(data))
        level = int(pipeline_config.get("generate_opt_level") or 0)
        if level == 0:
            if expand:
                data = expand_num_return_sequences(data, n)
            out = worker.generate(data)
            out.meta_info.setdefault("metrics", {})
            return out
        return self._generate_requests(data, worker, gc, pipeline_config)

    def join_idle_round(self, actor_cluster, pipeline_config) -> None:
        """Request-level mode across ranks is COLLECTIVE: every rank must enter every round.  The pipeline shards the samples with
        np.array_split sizes and walks its shard in batches, so ranks can own different numbers of batches (65 samples on 2 ranks at
        batch 32: two and one); a rank that has run out of batches joins the remaining rounds with no requests of its own and serves
        what the dispatcher hands it.  No-op below level 1 or with a single rank."""
        import torch.distributed as dist
        if int(pipeline_config.get("generate_opt_level") or 0) < 1 or not dist.is_initialized() or dist.get_world_size() == 1:
            return
        self._generate_requests(None, actor_cluster, {}, pipeline_config)

    def _generate_requests(self, data, worker, gc, pipeline_config) -> DataProto:
        import time
        import torch.distributed as dist
        B = len(data) if data is not None else 0
        self.round += 1
        reqs = [] if data is None else [DataProto(batch={k: v[i:i + 1] for k, v in data.batch.items()},
                          non_tensor_batch={k: v[i:i + 1] for k, v in data.non_tensor_batch.items()},
                          meta_info={"request_id": i, "generation_config": dict(gc, num_return_sequences=1)}) for i in range(B)]
        timeout = float(pipeline_config.get("rpc_timeout") or 3600)
        world = dist.get_world_size() if dist.is_initialized() else 1
        worker.start_server(DataProto(meta_info={}), request_complete_callback=self.report_response)
        if world > 1:
            # request-level dispatch over all ranks (reference generate_opt_level_1): collective -- every rank is in this call
            from socioreasoner_amd.dispatch import CrossRankDispatcher, default_store
            cap = int(pipeline_config.get("max_running_requests") or self.max_running_requests)
            disp = CrossRankDispatcher(default_store(), dist.get_rank(), world, self.round, max_running_requests=cap, timeout_s=timeout)
            try:
                got = disp.run(reqs, add_request=lambda r: worker.add_request(GenerateRequestType.ADD, r),
                               make_result_sink=lambda cb: setattr(self, "sink", cb),
                               make_request=lambda b, n, m: DataProto(batch=b, non_tensor_batch=n, meta_info=m),
                               alive_check=lambda: worker.add_request(GenerateRequestType.ALIVE_CHECK))
            finally:
                self.sink = None
                worker.stop_server()
            self.last_dispatch_stats = disp.stats
            self.results = {i: got[i] for i in range(B)}
            if data is None:
                return None
        else:
            self.results, self.expected = {}, B
            self.done.clear()
            for req in reqs:
                worker.add_request(GenerateRequestType.ADD, req)
            deadline = time.monotonic() + timeout
            while not self.done.wait(timeout=0.2):
                worker.add_request(GenerateRequestType.ALIVE_CHECK)         # raises if the server thread died
                if time.monotonic() > deadline:
                    worker.stop_server()
                    raise TimeoutError(f"{B - len(self.results)} of {B} generation requests did not complete")
            worker.stop_server()
        rows = [self.results[i] for i in range(B)]                      # re-sorted by prompt id (reference :293-294)
        return assemble_responses(data, rows, worker, pipeline_config)

    def open_stream(self, actor_cluster, pipeline_config) -> "RequestStream":
        """The request loop of level 1 kept OPEN across several batches of prompts (round 6; no reference counterpart -- the reference's two stages are
        two generate calls with the host flow between them): the two-stage pipeline adds stage-2 prompts while stage-1 prompts of other samples are
        still decoding, so the engine's rows never drain between the stages and the host flow runs under generation."""
        worker = actor_cluster
        gc = dict(worker.worker_config.generating_args or {})
        gc["num_return_sequences"] = 1
        gc.setdefault("max_new_tokens", int(pipeline_config.response_length))
        self.round += 1
        return RequestStream(worker, gc, float(pipeline_config.get("rpc_timeout") or 3600))


def assemble_responses(data: DataProto, rows: List[List[int]], worker, pipeline_config) -> DataProto:
    """token lists of the prompts of `data` (in its row order) -> the reference's 7 output tensors (postprocess_generate, reference :293-334)"""
    pad = worker.tokenizer.pad_token_id
    output_ids = hostops.gather_outputs_to_pad_tensor(rows, pad, device=data.batch["input_ids"].device)
    seq = hostops.concatenate_input_and_output(data.batch["input_ids"], output_ids, 1)
    out = hostops.postprocess_generate(prompts=data.batch, output=seq, num_return_sequences=1,
                                       sequence_length=int(pipeline_config.sequence_length),
                                       eos_token_id=worker.tokenizer.eos_token_id, pad_token_id=pad)
    return DataProto(batch=out, meta_info={"metrics": {}})


class RequestStream:
    """One generation server (ActorWorker.start_server: the strategy's request loop on its own thread) with requests added and answers collected while
    it runs.  `add` hands it one prompt row under a caller-chosen id, `collect` blocks for the next finished request(s)."""

    def __init__(self, worker, gc: Dict, timeout_s: float):
        import queue
        import time
        self.worker, self.gc = worker, gc
        self.answers: "queue.Queue" = queue.Queue()
        self.in_flight = 0
        self.deadline = time.monotonic() + timeout_s
        worker.start_server(DataProto(meta_info={}), request_complete_callback=self._report)

    def _report(self, data: DataProto):
        self.answers.put((int(data.meta_info["request_id"]), data.meta_info["output_token_ids"][0]))

    def add(self, request_ids: List[int], data: DataProto) -> None:
        """the rows of `data` as one request each.  The requests are built first and queued in one go: the server thread takes commands as fast as
        they arrive and starts its first admission when it finds the queue empty -- a producer slower than that would split its burst over several
        small admissions"""
        reqs = [DataProto(batch={k: v[i:i + 1] for k, v in data.batch.items()}, non_tensor_batch={k: v[i:i + 1] for k, v in data.non_tensor_batch.items()},
                          meta_info={"request_id": int(rid), "generation_config": dict(self.gc)}) for i, rid in enumerate(request_ids)]
        self.in_flight += len(reqs)
        for req in reqs:
            self.worker.add_request(GenerateRequestType.ADD, req)

    def collect(self, linger_s: float = 0.003) -> List[tuple]:
        """[(request id, tokens)] of the requests that have finished: waits for the first, then takes what arrives within `linger_s` of each other
        (rows that end in the same decode step are reported one after the other by the server thread)"""
        import queue
        import time
        got = []
        while not got:
            try:
                got.append(self.answers.get(timeout=0.2))
            except queue.Empty:
                self.worker.add_request(GenerateRequestType.ALIVE_CHECK)         # raises if the server thread died
                if time.monotonic() > self.deadline:
                    self.close()
                    raise TimeoutError(f"{self.in_flight} generation requests did not complete")
        while True:
            try:
                got.append(self.answers.get(timeout=linger_s))
            except queue.Empty:
                break
        self.in_flight -= len(got)
        return got

    def close(self) -> None:
        self.worker.stop_server()
