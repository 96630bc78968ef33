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

# SCRIPTED_OBJECTS=k (round 5): random weights emit no <answer>, so the faithful pipeline never gave seg_infer anything to do.  With k > 0 the run
# is unchanged up to the moment a response is DECODED -- the LM still generates its NEW_TOKENS tokens per stage on the engine -- and there the
# decoded text is replaced by a scripted answer of k objects (stage 1: k boxes; stage 2: the same boxes + two points each), a function of the row
# index only: SAM2 (Hiera-L, float32 unless SR_SAM2_DTYPE says otherwise) then encodes every satellite image and decodes k prompts per stage and
# sample, exactly the calls a real checkpoint's answers would cause (roll/pipeline/rlvr/seg_worker.py build_sam_prompts -> SegRasterStrategy.segment).
n_obj = int(os.environ.get("SCRIPTED_OBJECTS", 0))
n, new, out = int(os.environ.get("SOCIOSEG_NUM_SAMPLES", 64)), int(os.environ.get("NEW_TOKENS", 128)), os.environ.get("OUT", "/tmp/example_out")
os.environ["SOCIOSEG_NUM_SAMPLES"] = str(n)
os.environ.setdefault("SR_ALLOW_SYNTHETIC_WEIGHTS", "1")       # no SAM2 checkpoint offline: random weights on purpose (the provider refuses otherwise)
cfg = load_yaml_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "infer"), "rlvr_megatron")
cfg["response_length"] = new
cfg["actor_infer"]["generating_args"]["max_new_tokens"] = new
if os.environ.get("MAX_BATCH"):      # batch rows of the LM engine (the example YAML says 32 = BASELINE.json configs[2]; vLLM's own default admits 256 sequences, the reference's scheduler 128)
    cfg["actor_infer"]["strategy_args"]["strategy_config"]["max_batch"] = int(os.environ["MAX_BATCH"])
cfg["rollout_batch_size"] = int(os.environ.get("ROLLOUT_BATCH", min(n, 32)))      # (the reference YAML: 250 -- ROLLOUT_BATCH=250)
cfg["output_dir"] = out
cfg["logging_dir"] = os.path.join(out, "logs")
t0 = time.time()
init()
pipe = SocioSegInferPipeline(pipeline_config=SocioSegConfig.from_dict(cfg))
if n_obj > 0:
    import numpy as _np

    class Scripted:
        """the pipeline's tokenizer with batch_decode answering from a script (see above); everything else is the wrapped tokenizer's"""

        def __init__(self, tok):
            self._tok, self.stage = tok, 1

        def __getattr__(self, k):
            return getattr(self._tok, k)

        def batch_decode(self, ids, skip_special_tokens=False):
            outs = []
            for i in range(len(ids)):
                rng = _np.random.default_rng(1000 + i)
                objs = []
                for _ in range(n_obj):
                    x0, y0 = (int(v) for v in rng.integers(20, 480, 2))
                    b = [x0, y0, x0 + int(rng.integers(40, 240)), y0 + int(rng.integers(40, 240))]
                    o = {"bbox_2d": b}
                    if self.stage == 2:
                        o["points"] = [[(b[0] + b[2]) // 2, (b[1] + b[3]) // 2], [b[0] + 10, b[1] + 10]]
                    objs.append(o)
                outs.append(f"<think>scripted</think>\n<answer>{json.dumps(objs)}</answer>")
            return outs

    tok = Scripted(pipe.tokenizer)
    pipe.tokenizer = pipe.seg_infer.tokenizer = tok
    for stage, name in ((1, "segment_v4_map"), (2, "segment_v4_sat")):
        fn = getattr(pipe.seg_infer, name)
        setattr(pipe.seg_infer, name, (lambda fn, stage: lambda b: (setattr(tok, "stage", stage), fn(b))[1])(fn, stage))
t1 = time.time()
acc = pipe.run()
t2 = time.time()
files = {d: len(os.listdir(os.path.join(out, "result", d))) for d in ("stage1", "stage2", "render1", "render2") if os.path.isdir(os.path.join(out, "result", d))}
pred = getattr(pipe.seg_infer.strategy, "model", None)
print(json.dumps({"samples": n, "new_tokens_per_stage": new, "scripted_objects_per_stage": n_obj, "sam2_stats": getattr(pred, "stats", None), "streamed": getattr(pipe, "streamed", None),
                  "sam2_dtype": str(getattr(getattr(pred, "engine", None), "dt", None)), "build_s": round(t1 - t0, 1), "run_s": round(t2 - t1, 1), "samples_per_s": round(n / (t2 - t1), 2),
                  "giou_acc": acc, "files": files, "sam": type(pipe.seg_infer.strategy.model).__name__,
                  "wall_s_by_phase": {k: round(v, 2) for k, v in pipe.timing.items()},
                  "generate_calls": getattr(getattr(pipe.actor_infer, "strategy", None), "gen_stats", None)}))
