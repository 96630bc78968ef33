"""``create_strategy`` (reference: roll/distributed/strategy/factory.py:7-30).  The native engine registers as
``mi355x`` and also answers to ``vllm`` -- the name the shipped examples/infer/rlvr_megatron.yaml selects for the
``actor_infer`` role -- so that YAML runs unmodified.  ``seg_infer`` maps to the raster-tail strategy."""
from roll.distributed.strategy.strategy import InferenceStrategy


def create_strategy(worker) -> InferenceStrategy:
    strategy_name = worker.worker_config.strategy_args.strategy_name
    if strategy_name in ("mi355x", "vllm"):
        from roll.distributed.strategy.mi355x_strategy import Mi355xStrategy as strategy_cls
    elif strategy_name == "seg_infer":
        from roll.distributed.strategy.mi355x_strategy import SegRasterStrategy as strategy_cls
    else:
        raise ValueError(f"Unknown strategy name: {strategy_name} (this build implements the inference hot path only)")
    return strategy_cls(worker)
