"""Plugin boundary of the reference (roll/distributed/strategy/strategy.py:16-138): same class name, same method
names and argument meaning.  A strategy owns the engine of one worker (= one GPU / one DP rank)."""
from __future__ import annotations

from abc import ABC
from typing import Callable, Dict

import torch

from roll.distributed.scheduler.protocol import DataProto


class InferenceStrategy(ABC):
    strategy_name = None

    def __init__(self, worker):
        self.worker = worker
        self.model = None
        self.tokenizer = None
        self.worker_config = getattr(worker, "worker_config", None)
        self.model_update_comm_plan = {}

    def initialize(self, *args, **kwargs):
        raise NotImplementedError

    def forward_step(self, batch: DataProto, forward_func: Callable) -> Dict[str, torch.Tensor]:
        pass

    def get_data_input(self, batch: DataProto) -> DataProto:
        return batch

    def generate(self, *args, **kwargs):
        raise NotImplementedError

    def start_server(self, *args, **kwargs):
        raise NotImplementedError

    def add_request(self, command, data: DataProto, *args, **kwargs):
        raise NotImplementedError()

    def unwrap_model(self, *args, **kwargs):
        raise NotImplementedError

    def save_checkpoint(self, *args, **kwargs):
        raise NotImplementedError

    def load_checkpoint(self, *args, **kwargs):
        pass

    # trainer -> engine weight sync hooks (reference :66-83); the native engine loads weights itself
    def broadcast_parameter(self, src_pp_rank, dtype, shape, parameter_name):
        raise NotImplementedError

    def broadcast_bucket(self, src_pp_rank, meta_infos, bucket_size):
        raise NotImplementedError

    def update_parameter(self, parameter_name, weight, ranks_in_worker):
        raise NotImplementedError

    def update_parameter_in_bucket(self, meta_infos, buffer, ranks_in_worker):
        raise NotImplementedError

    def setup_collective_group(self, comm_plan, backend="nccl"):
        return None

    def load_states(self):
        raise NotImplementedError

    def offload_states(self, *args, **kwargs):
        raise NotImplementedError
