"""``SocioSegConfig`` (reference: roll/pipeline/rlvr/rlvr_config.py:79-324): same field names, read from the same
YAML; PPO / training fields are carried but unused."""
from roll.configs import Cfg, _wrap, parse_device_mapping


class SocioSegConfig(Cfg):
    @classmethod
    def from_dict(cls, data) -> "SocioSegConfig":
        c = cls(_wrap(dict(data)))
        for k, dflt in dict(seed=42, rollout_batch_size=250, prompt_length=4096, response_length=2048,
                            generate_opt_level=0, output_dir="./output/infer", rpc_timeout=3600).items():
            if c.get(k) is None:
                c[k] = dflt
        c["sequence_length"] = c["prompt_length"] + c["response_length"]
        for role in ("actor_infer", "seg_infer", "actor_train", "reference"):
            if c.get(role) is not None and c[role].get("device_mapping") is not None:
                c[role]["device_mapping"] = parse_device_mapping(c[role]["device_mapping"])
        return c
