"""Command-line entry of the two-stage SocioSeg inference run on the MI355X-native engine.

    python examples/start_rlvr_socioseg_pipeline_infer.py --config_path infer --config_name rlvr_megatron
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/start_rlvr_socioseg_pipeline_infer.py --config_path infer --config_name rlvr_megatron

The two flags are the reference launcher's (its hydra/omegaconf/dacite stack is replaced by the PyYAML loader in ``roll.configs``, which
reads the same YAML keys and resolves the same ``${...}`` interpolations); everything else -- process group, engine, pipeline -- is set up
from the file the flags point at.  Returns the run's gIoU as the process result line.
"""
import os
import sys
from argparse import ArgumentParser

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def cli() -> ArgumentParser:
    ap = ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--config_path", default="config", help="directory of the YAML file (relative to examples/ or to the working directory)")
    ap.add_argument("--config_name", default="sppo_config", help="YAML file name without its extension")
    ap.add_argument("--print_config", action="store_true", help="dump the resolved configuration before the run")
    return ap


def run(argv=None) -> float:
    opts = cli().parse_args(argv)
    from roll.configs import load_yaml_config
    from roll.distributed.scheduler.initialize import init
    from roll.pipeline.rlvr import rlvr_socioseg_vlm_pipeline_infer as infer
    tree = load_yaml_config(opts.config_path, opts.config_name)
    if opts.print_config:
        import yaml

        def plain(node):                                          # the loader's attribute-style mappings -> builtin containers
            if isinstance(node, dict):
                return {key: plain(val) for key, val in node.items()}
            return [plain(val) for val in node] if isinstance(node, (list, tuple)) else node
        print(yaml.safe_dump(plain(tree), sort_keys=False))
    init()                                                        # torch.distributed over RCCL when launched by torchrun; a no-op for one process
    job = infer.SocioSegInferPipeline(pipeline_config=infer.SocioSegConfig.from_dict(tree))
    score = job.run()
    print(f"[socioseg-infer] done: giou_acc={score}")
    return score


if __name__ == "__main__":
    run()
