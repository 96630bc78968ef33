"""CPU tests of the round-3 host logic: request-level dispatch ACROSS ranks (A15), world-size-8 sharding / exchange."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_DISPATCH_WORKER = r"""
import os, sys, time, queue
import numpy as np, torch
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
from socioreasoner_amd import dp, hostops
from roll.distributed.scheduler.generate_scheduler import GenerateScheduler
from roll.distributed.scheduler.protocol import DataProto
from roll.distributed.strategy.mi355x_strategy import Mi355xStrategy
from roll.pipeline.base_worker import ActorWorker
from socioreasoner_amd.config import geometry_tiny
from socioreasoner_amd.textproc import ByteTokenizer
from test_host_cpu import _tiny_cfg

rank, world, _ = dp.init_distributed("gloo")
tok = ByteTokenizer(geometry_tiny())
cfg = _tiny_cfg({tmp!r}, prompt_length=12, response_length=12)
DT = 0.02


class SlowFake(Mi355xStrategy):
    # the real request loop (start_server / add_request) over a scripted engine that takes DT seconds per generated token and serves
    # one request at a time: the answer (and so the time) depends on the prompt only, not on the rank that serves it
    max_batch = 1

    def initialize(self, model_provider=None):
        self.command_queue, self.tokenizer = queue.Queue(), tok
        self.busy_s = 0.0

    def generate(self, batch, generation_config):
        ids, mask = batch.batch["input_ids"], batch.batch["attention_mask"]
        rows = []
        for r, m in zip(ids, mask):
            p = r[m.bool()].tolist()
            n = p[0] % 16                                # answer length is written into the prompt's first token
            rows.append([(7 * t + 3) % 250 for t in (p * 4)[:n]] + [tok.eos_token_id])
        t = DT * sum(len(x) for x in rows)
        time.sleep(t)
        self.busy_s += t
        out = hostops.gather_outputs_to_pad_tensor(rows, generation_config["pad_token_id"], device=ids.device)
        return hostops.concatenate_input_and_output(ids, out, 1)


w = ActorWorker(cfg.actor_infer, cfg, rank, world, 0, "actor_infer")
w.strategy = SlowFake(w)
w.strategy.initialize()
w.tokenizer = tok
# skewed shards: rank 0 owns 8 prompts with 10-token answers, rank 1 owns 8 prompts with 1-token answers
n_own, alen = 8, (10 if rank == 0 else 1)
rng = np.random.default_rng(100 + rank)
ids = torch.full((n_own, 12), tok.pad_token_id, dtype=torch.long)
mask = torch.zeros(n_own, 12, dtype=torch.long)
for i in range(n_own):
    n = 4 + i % 5
    body = rng.integers(0, 250, n)
    body[0] = 16 * int(rng.integers(0, 10)) + alen
    ids[i, 12 - n:] = torch.from_numpy(body)
    mask[i, 12 - n:] = 1
pos = (mask.cumsum(-1) - 1).clamp(min=0)[:, None, :].repeat(1, 3, 1)
fresh = lambda: DataProto(batch={{"input_ids": ids.clone(), "attention_mask": mask.clone(), "position_ids": pos.clone()}}, non_tensor_batch={{}})
sched = GenerateScheduler()
cfg["generate_opt_level"] = 0
t0 = time.time(); out0 = sched.generate(fresh(), w, cfg); static_s = time.time() - t0         # static sharding: every rank serves its own shard
dp.barrier()
static_makespan = dp.all_reduce_max(static_s)

# ---- request-level dispatch across the ranks, work-conserving cap (one request in flight per worker = the fake engine's rows)
cfg["generate_opt_level"] = 1
cfg["max_running_requests"] = 1
w.strategy.busy_s = 0.0
dp.barrier()
t0 = time.time(); out1 = sched.generate(fresh(), w, cfg); dyn_s = time.time() - t0
st1 = dict(sched.last_dispatch_stats)
busy1 = w.strategy.busy_s
dyn_makespan = dp.all_reduce_max(dyn_s)
for k in out0.batch:
    assert torch.equal(out1.batch[k], out0.batch[k]), (rank, k)       # every rank gets ITS prompts' answers, in prompt order
if rank == 1:
    assert st1["served_for_others"] >= 3, st1                          # the lightly loaded rank took over long requests of rank 0
if rank == 0:
    assert st1["shipped_out"] >= 3 and sum(st1["sent_to"]) == 2 * n_own, st1
other_busy = dp.all_reduce_max(busy1)
assert dyn_makespan < 0.9 * static_makespan, (dyn_makespan, static_makespan)
# the two engines were busy for about the same time (static: 8 x 11 x DT against 8 x 2 x DT)
tb = torch.tensor([busy1], dtype=torch.float64)
import torch.distributed as dist
both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
dist.all_gather(both, tb)
b0, b1 = float(both[0]), float(both[1])
assert min(b0, b1) > 0.55 * max(b0, b1), (b0, b1)

# ---- the reference's cap (128): a batch this small is dealt out at once, interleaved over the ranks (reference :180-187)
cfg["max_running_requests"] = 128
dp.barrier()
out2 = sched.generate(fresh(), w, cfg)
st2 = dict(sched.last_dispatch_stats)
for k in out0.batch:
    assert torch.equal(out2.batch[k], out0.batch[k]), (rank, k)
if rank == 0:
    assert st2["sent_to"] == [n_own, n_own], st2
dp.barrier()
print("ok", rank, "static %.2fs dynamic %.2fs busy %.2f / %.2f" % (static_makespan, dyn_makespan, b0, b1))
"""


def test_request_level_dispatch_across_two_ranks_gloo(tmp_path):
    """A15 across ranks: with skewed answer lengths the least-loaded dispatch (reference generate_scheduler.py:180-187) moves
    requests from the loaded rank to the idle one -- same answers as the static sharding, in prompt order, shorter makespan,
    both engines busy about equally long -- and with the reference's cap of 128 it deals a small batch out interleaved."""
    script = tmp_path / "w.py"
    script.write_text(_DISPATCH_WORKER.format(root=ROOT, tmp=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") == 2


_WORLD8_WORKER = r"""
import os, sys, torch
sys.path.insert(0, {root!r})
from socioreasoner_amd import dp
rank, world, _ = dp.init_distributed("gloo")
assert world == 8
# BASELINE.json configs[3]: 256 tiles over 8 ranks = 32 per rank, contiguous, rank order
assert dp.split_sizes(256, 8) == [32] * 8 and dp.shard_range(256, rank, 8) == (32 * rank, 32 * rank + 32)
a, b = dp.shard_range(256, rank, world)
res = torch.arange(a, b, dtype=torch.int64).unsqueeze(1) * 1000 + torch.arange(130)       # the bench's result row: 128 tokens + 2 IoU counts
full = dp.all_gather_rows(res, 256)
assert full.shape == (256, 130) and torch.equal(full, torch.arange(256).unsqueeze(1) * 1000 + torch.arange(130))
again = dp.all_gather_rows(res + 1, 256)
assert torch.equal(full, torch.arange(256).unsqueeze(1) * 1000 + torch.arange(130)), "an earlier result must not alias the exchange buffer"
# ragged: 250 samples (the reference's rollout_batch_size) -> 32, 32, 31, ...
n = 250
a, b = dp.shard_range(n, rank, world)
full = dp.all_gather_rows(torch.arange(a, b, dtype=torch.int64).unsqueeze(1), n)
assert full[:, 0].tolist() == list(range(n))
info = dp.exchange_info()
assert info["nranks"] == 8
assert dp.all_reduce_max(float(rank)) == 7.0
dp.barrier()
print("ok", rank)
"""


def test_world_size_8_sharding_and_exchange_gloo(tmp_path):
    """configs[3]'s layout on 8 ranks (gloo on CPU): split sizes, result-row all-gather in rank order, ragged sample counts."""
    script = tmp_path / "w8.py"
    script.write_text(_WORLD8_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                          "--master-addr", "127.0.0.1", "--master-port", "29543", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") == 8


def _to_sam2_package_names(sd):
    """HF Sam2Model names -> the sam2 package's (the two public module trees; hand-written inverse of socioreasoner_amd.sam2's table)"""
    import torch
    out = {}
    for k, v in sd.items():
        n = k
        if n == "no_memory_embedding":
            n = "no_mem_embed"
        elif n.startswith("vision_encoder.backbone."):
            n = "image_encoder.trunk." + n[len("vision_encoder.backbone."):]
            n = n.replace("patch_embed.projection.", "patch_embed.proj.").replace(".layer_norm1.", ".norm1.").replace(".layer_norm2.", ".norm2.")
            n = n.replace(".mlp.proj_in.", ".mlp.layers.0.").replace(".mlp.proj_out.", ".mlp.layers.1.")
        elif n.startswith("vision_encoder.neck.convs."):
            n = "image_encoder.neck.convs." + n[len("vision_encoder.neck.convs."):].replace(".weight", ".conv.weight").replace(".bias", ".conv.bias")
        elif n == "prompt_encoder.shared_embedding.positional_embedding":
            n = "sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"
        elif n == "prompt_encoder.point_embed.weight":
            for i in range(v.shape[0]):
                out[f"sam_prompt_encoder.point_embeddings.{i}.weight"] = v[i:i + 1].clone()
            continue
        elif n.startswith("prompt_encoder."):
            n = "sam_" + n
        elif n.startswith("mask_decoder."):
            n = "sam_" + n
            for a, b in ((".layer_norm_final_attn.", ".norm_final_attn."), (".layer_norm1.", ".norm1."), (".layer_norm2.", ".norm2."), (".layer_norm3.", ".norm3."),
                         (".layer_norm4.", ".norm4."), (".o_proj.", ".out_proj."), ("upscale_conv1.", "output_upscaling.0."), ("upscale_layer_norm.", "output_upscaling.1."),
                         ("upscale_conv2.", "output_upscaling.3.")):
                n = n.replace(a, b)
            if "hypernetworks" in n or "iou_prediction_head" in n:
                n = n.replace(".layers.0.", ".layers.1.").replace(".proj_in.", ".layers.0.").replace(".proj_out.", ".layers.2.")
            else:
                n = n.replace(".mlp.proj_in.", ".mlp.layers.0.").replace(".mlp.proj_out.", ".mlp.layers.1.")
        out[n] = v
    # what the real checkpoint also carries and the image path never reads
    for extra in ("memory_encoder.fuser.layers.0.gamma", "memory_attention.layers.0.norm1.weight", "maskmem_tpos_enc", "obj_ptr_proj.layers.0.weight", "no_mem_pos_enc",
                  "no_obj_ptr", "mask_downsample.weight", "sam_prompt_encoder.mask_downscaling.0.weight", "sam_mask_decoder.pred_obj_score_head.layers.0.weight"):
        out[extra] = torch.zeros(1)
    return out


def test_sam2_package_checkpoint_names_map_onto_the_names_the_engine_loads():
    """roll.models.model_providers.sam2_seg_model_provider renames a sam2-package checkpoint (model_providers.py:540-541 of the reference loads
    sam2_hiera_large.pt) before Sam2Engine.load_state_dict: every parameter the engine expects must come out, with its tensor, and the
    memory / video modules must be dropped."""
    import torch
    from socioreasoner_amd import sam2
    g = sam2.Sam2Geometry()
    names = sam2.param_shapes(g)
    sd = {n: torch.full((1,) if not shp else tuple(min(int(d), 4) for d in shp), float(i)) for i, (n, shp, _, _) in enumerate(names)}
    sd["prompt_encoder.point_embed.weight"] = torch.arange(4 * 3, dtype=torch.float32).reshape(4, 3)
    pkg = _to_sam2_package_names(sd)
    assert not any(k.startswith(("vision_encoder.", "prompt_encoder.", "mask_decoder.")) for k in pkg)
    back = sam2.rename_sam2_checkpoint(pkg)
    missing = sorted(set(sd) - set(back))
    assert not missing, missing[:8]
    for k in sd:
        assert torch.equal(back[k], sd[k]), k
    extra = set(back) - set(sd)
    assert extra <= {"mask_decoder.pred_obj_score_head.proj_in.weight"}, sorted(extra)[:8]      # (object-score head: loaded by HF, unused by predict)


def test_seg_infer_provider_default_follows_the_configuration():
    """seg_worker.py:540 of the reference hands sam2_seg_model_provider to the strategy; here a configuration that names no SAM2 model keeps the
    offline stand-in (host-flow tests), one that names a model goes to the MI355X SAM2 provider -- which refuses to run without a GPU
    instead of falling back to anything."""
    import pytest
    import torch
    from roll.pipeline.rlvr import rlvr_socioseg_vlm_pipeline_infer as P
    from socioreasoner_amd import socioseg_data
    assert isinstance(P._default_sam_provider(model_args={}), socioseg_data.SyntheticSamPredictor)
    assert isinstance(P._default_sam_provider(model_args=None), socioseg_data.SyntheticSamPredictor)
    if not torch.cuda.is_available():
        with pytest.raises(Exception) as ei:
            P._default_sam_provider(model_args={"model_name_or_path": "synthetic:sam2-tiny"})
        assert "GPU" in str(ei.value) or "cuda" in str(ei.value).lower() or "HIP" in str(ei.value)
