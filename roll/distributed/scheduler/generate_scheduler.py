"""``GenerateScheduler`` (reference: roll/distributed/scheduler/generate_scheduler.py:63-334).

The reference fans a batch out to its DP workers through Ray (level 0) or dispatches single requests to the least
loaded worker and collects them through callbacks (level 1).  Here every torchrun rank already holds its shard of the
batch (socioreasoner_amd.dp), so the scheduler drives the LOCAL worker: level 0 is one ``generate`` call, level 1
splits the batch into single-prompt requests served by the worker's request loop and re-assembled in prompt order."""
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
    def __init__(self):
        self.lock = threading.Lock()
        self.results: Dict[int, List[int]] = {}
        self.done = threading.Event()

    def report_response(self, data: DataProto):
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

    def _generate_requests(self, data: DataProto, worker, gc, pipeline_config) -> DataProto:
        B = len(data)
        self.results, self.expected = {}, B
        self.done.clear()
        worker.start_server(DataProto(meta_info={}), request_complete_callback=self.report_response)
        for i in range(B):
            req = DataProto(batch={k: v[i:i + 1] for k, v in data.batch.items()},
                            non_tensor_batch={k: v[i:i + 1] for k, v in data.non_tensor_batch.items()},
                            meta_info={"request_id": i, "generation_config": dict(gc, num_return_sequences=1)})
            worker.add_request(GenerateRequestType.ADD, req)
        import time
        deadline = time.monotonic() + float(pipeline_config.get("rpc_timeout") or 3600)
        while not self.done.wait(timeout=0.2):
            worker.add_request(GenerateRequestType.ALIVE_CHECK)         # raises if the server thread died
            if time.monotonic() > deadline:
                worker.stop_server()
                raise TimeoutError(f"{B - len(self.results)} of {B} generation requests did not complete")
        worker.stop_server()
        pad = worker.tokenizer.pad_token_id
        rows = [self.results[i] for i in range(B)]                      # re-sorted by prompt id (reference :293-294)
        output_ids = hostops.gather_outputs_to_pad_tensor(rows, pad, device=data.batch["input_ids"].device)
        seq = hostops.concatenate_input_and_output(data.batch["input_ids"], output_ids, 1)
        out = hostops.postprocess_generate(prompts=data.batch, output=seq, num_return_sequences=1,
                                           sequence_length=int(pipeline_config.sequence_length),
                                           eos_token_id=worker.tokenizer.eos_token_id, pad_token_id=pad)
        return DataProto(batch=out, meta_info={"metrics": {}})
