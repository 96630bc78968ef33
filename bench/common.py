"""Constants, arguments and helpers shared by the bench modules."""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16
VIT_GFLOP, PREFILL_GFLOP = 1342.9, 2516.3      # algorithmic work per 448 x 448 tile
N_NEW = 128
RAGGED_LO, RAGGED_HI = 64, 192      # ragged phase: per-request max_new uniform in [64, 192] (mean 128)
KV_BYTES_PER_TOKEN = 36864.0        # K + V of one cached token of one sequence, all 36 layers


def lm_weight_bytes(g):
    """(bytes of the 36 layers' linears, bytes of the tied LM head) in bf16: what one decode step streams once, whatever the batch."""
    t = g.text
    qn = (t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim
    per_layer = (qn * t.hidden_size + t.hidden_size * t.num_attention_heads * t.head_dim + 3 * t.intermediate_size * t.hidden_size) * 2
    return per_layer * t.num_hidden_layers, t.vocab_size * t.hidden_size * 2


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="batch rows per GPU (32 = configs[2], the default; 1 = configs[1])")
    ap.add_argument("--static", action="store_true", help="one static batch of --batch tiles per step instead of continuous batching")
    ap.add_argument("--continuous", action="store_true", help="(default for --batch > 1) serve --waves x batch requests through the scheduler")
    ap.add_argument("--tile", type=int, default=448, choices=[448, 896],
                    help="tile edge in pixels (896 = BASELINE.json configs[4]'s high-res tiles: 4096 patches, 1024 image tokens, 1216-token prompt); "
                         "the MFMA fractions are only quoted for 448")
    ap.add_argument("--pair", action="store_true", help="reference-faithful sample: TWO images (map + satellite tile) per request, 706-token prompt "
                    "(SURVEY.md section 8(D) 'reported separately'); value is then samples/s")
    ap.add_argument("--waves", type=int, default=4, help="continuous mode: a step serves waves x batch tile requests through the batch rows")
    ap.add_argument("--admit-cus", default="auto", help="CUs per shader engine (of 8) given to the overlapped admission stream, or auto (chosen per admission)")
    ap.add_argument("--no-overlap", action="store_true", help="continuous mode: admit between decode steps on one stream (round-1 behaviour) instead of "
                    "staging the next admission on a CU-masked stream under the running rows' decode")
    ap.add_argument("--drain", action="store_true", help="continuous mode: drain the batch rows between steps (every step starts with an exposed admission "
                    "on an idle engine, rounds 1-3) instead of serving the steps' requests as ONE stream")
    ap.add_argument("--poll", type=int, default=8, help="continuous mode: decode steps queued per scheduling round (rows are released / admitted between "
                    "rounds; 8 = ContinuousBatcher's own default.  Round 6, same box, alternating: 82.3-83.0 tiles/s at 16, 83.0 at 8, 82.8 at 4 -- the rows "
                    "return to the whole chip at the first poll after an admission has landed)")
    ap.add_argument("--poll-ragged", type=int, default=4, help="the same for the ragged phase (0 = --poll): rows that end on different steps are refilled "
                    "sooner "
                    "with short rounds -- measured 59.8 / 60.8 / 62.1 tiles/s at 16 / 8 / 4")
    ap.add_argument("--no-latency", action="store_true", help="skip the side measurements (static batch, drained step, ragged, batch 1, ...)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sam", action="store_true", help="skip the SAM2 (seg_infer) timing")
    ap.add_argument("--no-more-rows", action="store_true", help="skip the 64- and 128-row points (child processes)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the two-stage pipeline timing with SAM2 at work (a child process)")
    ap.add_argument("--pipeline-ab", action="store_true",
                    help="also time the pipeline in the reference's batch order (SOCIOSEG_STREAM=0): one more child process, ~35 s")
    ap.add_argument("--no-pmc", action="store_true", help="no rocprofv3 child passes in this run: neither the in-situ kernel trace of the decode weight stream "
                    "(roofline.avg_launch_us then comes from the launch-only replay, and says so) nor the FETCH_SIZE / WRITE_SIZE passes")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="BASELINE.json configs[4] weights: LM decoder linears fp8 e4m3, per-channel scale")
    ap.add_argument("--fp8-mx", action="store_true", help="--fp8 plus MX fp8 activations in the prefill linears (fp8 x fp8 block-scaled MFMA, "
                    "lm_weight_dtype 2)")
    ap.add_argument("--gather-logits", action="store_true",
                    help="verification mode (static batch): all-gather the float32 logits of every decode step (north_star's literal exchange)")
    ap.add_argument("--pipeline-samples", type=int, default=250, help="samples of the two-stage pipeline leg (the reference's rollout_batch_size is 250)")
    args = ap.parse_args(argv)
    if args.fp8_mx:
        args.fp8 = True
    return args


def child_env(**extra):
    """Environment of a child process that must run as ONE rank of its own (no torchrun variables)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK")}
    env.update(extra)
    return env


def last_json_line(text):
    import json
    return json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])


def clean(o):
    """NaN (fractions that are not quoted for this workload) -> null."""
    if isinstance(o, float) and o != o:
        return None
    if isinstance(o, dict):
        return {k: clean(v) for k, v in o.items()}
    if isinstance(o, list):
        return [clean(v) for v in o]
    return o


def frac_of_mfma_peak(gflop, ms):
    return round(gflop / (ms * 1e-3) / 1e3 / MFMA_PEAK_TFLOPS, 4) if ms > 0 else None


PYTHON = sys.executable
