"""``ActorWorker`` for the infer path (reference: roll/pipeline/base_worker.py:107-204, 343-382).

One worker = one torchrun rank = one MI355X (the reference creates them as Ray actors, roll/distributed/executor).
``generate`` injects eos / pad into the generation config, calls the strategy, and lays the result out with
``postprocess_generate`` exactly as the reference does (7 tensors, right-padded, mRoPE ids extended by +1 per token).
There is no load/offload of model state around the call: the weights stay resident in HBM (288 GB per GPU).
"""
from __future__ import annotations

import logging
import threading

import torch

from roll.distributed.scheduler.protocol import DataProto
from roll.distributed.strategy.factory import create_strategy
from socioreasoner_amd import hostops

logger = logging.getLogger("roll.worker")


class RankInfo:
    def __init__(self, rank=0, world_size=1, local_rank=0):
        self.dp_rank, self.dp_size, self.local_rank = rank, world_size, local_rank
        self.tp_rank = self.pp_rank = self.cp_rank = 0


class Worker:
    """What a strategy sees of its worker (reference: roll/distributed/executor/worker.py:41-204)."""

    def __init__(self, worker_config, pipeline_config=None, rank=0, world_size=1, local_rank=0, cluster_name=None):
        self.worker_config, self.pipeline_config = worker_config, pipeline_config
        self.rank, self.world_size = rank, world_size
        self.rank_info = RankInfo(rank, world_size, local_rank)
        self.cluster_name = cluster_name or (worker_config.get("name") if hasattr(worker_config, "get") else None) or "worker"
        self.worker_name = f"{self.cluster_name}-{rank}"
        self.strategy = None
        self.tokenizer = None

    def initialize(self, pipeline_config=None, model_provider=None, tokenizer=None):
        if pipeline_config is not None:
            self.pipeline_config = pipeline_config
        self.strategy = create_strategy(self)
        self.strategy.initialize(model_provider)
        self.tokenizer = tokenizer if tokenizer is not None else getattr(self.strategy, "tokenizer", None)
        return self


class ActorWorker(Worker):
    @torch.no_grad()
    def generate(self, data: DataProto) -> DataProto:
        ga = self.worker_config.generating_args if hasattr(self.worker_config, "generating_args") else None
        if "generation_config" in data.meta_info:
            generation_config = dict(data.meta_info["generation_config"])
        else:
            generation_config = dict(ga or {})
        generation_config.setdefault("num_return_sequences", 1)
        generation_config.setdefault("num_beams", 1)
        generation_config.setdefault("repetition_penalty", 1.0)
        generation_config["eos_token_id"] = [self.tokenizer.eos_token_id] + list(getattr(self.tokenizer, "additional_special_tokens_ids", []) or [])
        generation_config["pad_token_id"] = self.tokenizer.pad_token_id
        logger.info("%s generate global step %s", self.worker_name, data.meta_info.get("global_step", 0))
        output = self.strategy.generate(batch=data, generation_config=generation_config)
        out = hostops.postprocess_generate(
            prompts=data.batch, output=output, num_return_sequences=int(generation_config["num_return_sequences"]),
            sequence_length=int(self.pipeline_config.sequence_length), eos_token_id=self.tokenizer.eos_token_id,
            pad_token_id=self.tokenizer.pad_token_id)
        return DataProto(batch={k: v.cpu() for k, v in out.items()}, meta_info={"metrics": {}})

    # ---- request-level serving (generate_opt_level 1): the strategy's loop runs on its own thread, commands arrive
    # through its queue (reference base_worker.py:162-204, 343-382)
    def start_server(self, data: DataProto, request_complete_callback=None):
        cb = request_complete_callback or data.meta_info.get("response_callback_fn")
        self.server_error = None

        def loop():
            try:
                # the HIP current device is per THREAD and defaults to 0: without this, every launch / copy / graph capture of
                # a worker with local_rank > 0 would target device 0 against a workspace that lives on device N
                eng = getattr(self.strategy, "engine", None)
                if eng is not None and getattr(eng, "device", None) is not None:
                    import torch
                    torch.cuda.set_device(eng.device)
                self.strategy.start_server(data=data, request_complete_callback=cb)
            except BaseException as e:  # noqa: BLE001  (surfaced by ALIVE_CHECK instead of dying silently)
                self.server_error = e

        self.thread_server = threading.Thread(target=loop, daemon=True)
        self.thread_server.start()

    def add_request(self, command, data: DataProto = None):
        from roll.utils.functionals import GenerateRequestType
        if command == GenerateRequestType.ALIVE_CHECK:
            if self.server_error is not None or not self.thread_server.is_alive():
                raise RuntimeError("generation server stopped unexpectedly") from self.server_error
            return DataProto(meta_info={})
        if command == GenerateRequestType.ADD:
            gc = dict(data.meta_info.get("generation_config") or self.worker_config.generating_args or {})
            gc["eos_token_id"] = [self.tokenizer.eos_token_id] + list(getattr(self.tokenizer, "additional_special_tokens_ids", []) or [])
            gc["pad_token_id"] = self.tokenizer.pad_token_id
            data.meta_info["generation_config"] = gc
        self.strategy.add_request(command, data)
        return DataProto(meta_info={})

    def stop_server(self):
        from roll.utils.functionals import GenerateRequestType
        self.strategy.add_request(GenerateRequestType.STOP, None)
        self.thread_server.join(timeout=60)
