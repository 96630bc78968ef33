"""Assembly of the ONE JSON line the driver parses (contract: the round prompt / DESIGN.md section 5)."""
from __future__ import annotations

import torch

from .common import N_NEW, PREFILL_GFLOP, VIT_GFLOP, frac_of_mfma_peak


def exchange_report(args, world):
    """What the communicator says about itself.  An N-rank record verifies itself: the group really has N ranks (hard error otherwise -- a
    scaling number from a degenerate group would be worthless), and what RCCL logged about its transports is reported as checks, not guessed."""
    from socioreasoner_amd import dp
    ex = dp.exchange_info()
    if world > 1:
        assert ex["nranks"] == world == args.gpus, ex
        if ex.get("backend") == "nccl":
            ex["checks"] = {"rccl_logged_nranks_eq_world": ex.get("log_nranks") == [world], "channels_via_p2p_xgmi": ex.get("channels_via_p2p", 0) > 0,
                            "no_channel_via_net": ex.get("channels_via_net", 0) == 0,
                            "no_host_staging": True}      # all_gather_into_tensor on device buffers (dp._gather_equal)
            ex["verified"] = all(ex["checks"].values())
    ex["payload"] = ("float32 logits all-gather per decode step (verification mode)" if args.gather_logits
                     else "one all-gather of 1 KB result rows per tile and step")
    return ex


def phases(wl):
    """Per-step phase times of the timed region + the scheduler's counters + the MFMA fractions of the admissions that had the whole chip."""
    a, B, pm, sc = wl.args, wl.B, wl.phase_ms, wl.sched
    out = {k: round(v / a.steps, 3) for k, v in pm.items()}
    if wl.continuous:
        rounds = max(sc["rounds"], 1)
        out["scheduler"] = dict(
            {k: v // a.steps for k, v in sc.items() if k not in ("host_ms", "poll_wait_ms", "share_model")}, overlap=wl.overlap, admit_cus_per_se=wl.shares,
            share_model=sc.get("share_model"), host_ms_per_round=round(sc["host_ms"] / rounds, 3), poll_wait_ms_per_round=round(sc["poll_wait_ms"] / rounds, 3),
            decode_step_ms_shared=round(pm.get("decode_shared", 0.0) / max(sc["steps_shared"], 1), 4),
            note=(f"spans named *_shared ran concurrently on disjoint CU sets (admission share of the CUs: {a.admit_cus} of 8 per shader engine): "
                  "they do not add up to ms_per_step") if wl.overlap else None)
    # batches of B tiles inside the timed region whose admission had the whole chip (the MFMA fractions are quoted on those)
    per = (sc["admitted"] - sc["staged_shared"]) / B if wl.continuous else a.steps
    vit_ms, pre_ms = (pm["vit"] / per, pm["prefill"] / per) if per > 0 else (0.0, 0.0)
    q = wl.quotes_mfma
    out["vit_mfma_frac"] = frac_of_mfma_peak(VIT_GFLOP * B, vit_ms) if q else None
    out["prefill_mfma_frac"] = frac_of_mfma_peak(PREFILL_GFLOP * B, pre_ms) if q else None
    out["forward_mfma_frac"] = frac_of_mfma_peak((VIT_GFLOP + PREFILL_GFLOP) * B, vit_ms + pre_ms) if q else None
    return out


def decode_step_alone_ms(wl):
    """One decode step that had the chip to itself inside the timed region (continuous mode: steps that shared it with an admission are listed apart)."""
    if not wl.continuous:
        return wl.phase_ms["decode"] / (wl.args.steps * (N_NEW - 1))
    return wl.phase_ms["decode"] / max(wl.sched["steps"] - wl.sched["steps_shared"], 1)


def workload_text(wl):
    a, B = wl.args, wl.B
    if wl.continuous:
        how = (f"continuous batching ({'next admission overlapped with decode' if wl.overlap else 'admit on finish'}): {wl.n_req} tile requests per step "
               f"through {B} rows" + (" (rows drained between steps), " if a.drain else f", the {a.steps} steps served as one request stream, "))
    else:
        how = "static batch, "
    cfg = None
    if a.tile == 448 and not a.fp8 and not a.pair:
        cfg = "BASELINE.json configs[2]" if B == 32 and wl.continuous else "BASELINE.json configs[1]" if B == 1 else None
    elif a.tile == 896 and a.fp8:
        cfg = "BASELINE.json configs[4], one GPU's share"
    return (f"SocioReasoner-3B {'fp8-weight' if a.fp8 else 'bf16'}, {B} batch row(s)/GPU, " + how
            + f"{wl.n_img} x {a.tile}x{a.tile} synthetic image(s) per request, {wl.s_prompt}-token prompt, greedy decode of {N_NEW} tokens (EOS ignored), "
              "raster tail; "
              "random-init weights (counter-based generator, seed 0)" + (f" [{cfg}]" if cfg else ""))


def assemble(wl, dt, res, roof, cpu, cpu_hf, side):
    a, world = wl.args, wl.world
    dtype = "bf16"
    if a.fp8:
        dtype = ("fp8-e4m3 LM linears: prefill fp8 x fp8 on the block-scaled MFMA (MX activations), decode fp8 weights x bf16 activations" if a.fp8_mx
                 else "bf16 (fp8-e4m3 LM linear weights, bf16 activations / MFMA)")
    if wl.continuous:
        sched = ("continuous batching through B rows; the next requests' ViT + prefill are staged into spare KV slots on a CU-masked stream under the "
                 "running rows' decode") if wl.overlap else "continuous batching (admit on finish) through B rows"
    else:
        sched = "static batch"
    out = {
        "metric": ("satellite tiles/sec (448x448, SocioReasoner-3B)" if not a.pair
                   else "reference-faithful samples/sec (map + satellite tile per sample, SocioReasoner-3B)"),
        "value": round(world * wl.n_req * a.steps / dt, 4), "unit": "tiles/s" if not a.pair else "samples/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": workload_text(wl), "tiles_per_gpu_per_step": wl.n_req, "scheduling": sched, "parallelism": f"dp{world}",
                   "decode": "hipGraph" if not a.no_graph else "eager", "exchange": exchange_report(a, world)},
        "roofline": roof, "cpu_baseline": cpu, "cpu_baseline_torch_bf16": cpu_hf, "phase_ms_per_step": phases(wl),
    }
    out.update(side)
    out.update({"weights_load_s": round(wl.load_s, 1), "workspace_GB": round(wl.eng.workspace_bytes / 1e9, 2),
                "result_checksum": int(res.sum().item()),
                # one per tile of the last step, in tile order over all ranks
                "result_row_checksums": [int(v) for v in res.sum(dim=1).tolist()] if res.shape[0] <= 256 else None,
                "host_threads_per_rank": torch.get_num_threads()})
    return out
