"""examples/infer/rlvr_megatron.yaml end to end on one GPU at a size that finishes in a minute: the YAML as shipped (3B geometry, SAM2 Hiera-L,
both with synthetic weights -- no checkpoint offline), SOCIOSEG_NUM_SAMPLES synthetic samples, response_length cut to N new tokens.
Random weights emit no <answer>, so seg_infer sees empty prompts: this run shows the drop-in path works and what the host flow costs; the
SAM2 forward inside the pipeline is exercised by tests/test_gpu_sam2.py::test_pipeline_two_stage_flow_with_sam2_on_device."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roll.configs import load_yaml_config  # noqa: E402
from roll.distributed.scheduler.initialize import init  # noqa: E402
from roll.pipeline.rlvr.rlvr_socioseg_vlm_pipeline_infer import SocioSegConfig, SocioSegInferPipeline  # noqa: E402

n, new, out = int(os.environ.get("SOCIOSEG_NUM_SAMPLES", 64)), int(os.environ.get("NEW_TOKENS", 128)), os.environ.get("OUT", "/tmp/example_out")
os.environ["SOCIOSEG_NUM_SAMPLES"] = str(n)
os.environ.setdefault("SR_ALLOW_SYNTHETIC_WEIGHTS", "1")       # no SAM2 checkpoint offline: random weights on purpose (the provider refuses otherwise)
cfg = load_yaml_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "infer"), "rlvr_megatron")
cfg["response_length"] = new
cfg["actor_infer"]["generating_args"]["max_new_tokens"] = new
cfg["rollout_batch_size"] = min(n, 32)
cfg["output_dir"] = out
cfg["logging_dir"] = os.path.join(out, "logs")
t0 = time.time()
init()
pipe = SocioSegInferPipeline(pipeline_config=SocioSegConfig.from_dict(cfg))
t1 = time.time()
acc = pipe.run()
t2 = time.time()
files = {d: len(os.listdir(os.path.join(out, "result", d))) for d in ("stage1", "stage2", "render1", "render2") if os.path.isdir(os.path.join(out, "result", d))}
print(json.dumps({"samples": n, "new_tokens_per_stage": new, "build_s": round(t1 - t0, 1), "run_s": round(t2 - t1, 1), "samples_per_s": round(n / (t2 - t1), 2),
                  "giou_acc": acc, "files": files, "sam": type(pipe.seg_infer.strategy.model).__name__,
                  "wall_s_by_phase": {k: round(v, 2) for k, v in pipe.timing.items()}}))
