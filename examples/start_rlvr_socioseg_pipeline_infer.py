"""Launcher with the reference's CLI (examples/start_rlvr_socioseg_pipeline_infer.py:12-39):
    python examples/start_rlvr_socioseg_pipeline_infer.py --config_path infer --config_name rlvr_megatron
hydra/omegaconf/dacite are replaced by the PyYAML loader in roll.configs (same YAML keys)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from roll.configs import load_yaml_config  # noqa: E402
from roll.distributed.scheduler.initialize import init  # noqa: E402
from roll.pipeline.rlvr.rlvr_socioseg_vlm_pipeline_infer import SocioSegConfig, SocioSegInferPipeline  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--config_path", help="The path of the main configuration file", default="config")
    parser.add_argument("--config_name", help="The name of the main configuration file (without extension).", default="sppo_config")
    args = parser.parse_args()
    cfg = load_yaml_config(args.config_path, args.config_name)
    ppo_config = SocioSegConfig.from_dict(cfg)
    init()
    pipeline = SocioSegInferPipeline(pipeline_config=ppo_config)
    pipeline.run()


if __name__ == "__main__":
    main()
