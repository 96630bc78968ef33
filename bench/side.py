"""Side measurements carried in the same JSON line (rank 0 of an N = 1 run only; none of them is inside the timed region of `value`)."""
from __future__ import annotations

import os
import subprocess
import time

import numpy as np
import torch

from .common import N_NEW, PREFILL_GFLOP, PYTHON, RAGGED_HI, RAGGED_LO, ROOT, VIT_GFLOP, child_env, frac_of_mfma_peak, last_json_line


def static_batch(wl):
    """The same kernels as ONE static batch of B tiles (no scheduler, whole chip, warm): the phase times the MFMA fractions are quoted on."""
    B, st = wl.B, {"vit": 0.0, "prefill": 0.0, "decode": 0.0, "raster": 0.0}
    wl.step_static(B, gather=False)          # (rank 0 only: no collective in here)
    for _ in range(2):
        wl.step_static(B, st, gather=False)
    q = wl.quotes_mfma
    return {"workload": f"one static batch of {B} tiles per step (same engine, same kernels, no scheduler)", "steps": 2,
            "phase_ms": {k: round(v / 2, 3) for k, v in st.items()}, "decode_step_ms": round(st["decode"] / 2 / (N_NEW - 1), 4),
            "vit_mfma_frac": frac_of_mfma_peak(VIT_GFLOP * B, st["vit"] / 2) if q else None,
            "prefill_mfma_frac": frac_of_mfma_peak(PREFILL_GFLOP * B, st["prefill"] / 2) if q else None,
            "forward_mfma_frac": frac_of_mfma_peak((VIT_GFLOP + PREFILL_GFLOP) * B, (st["vit"] + st["prefill"]) / 2) if q else None}


def drained_step(wl, n_dr=2):
    """The same step on a DRAINED engine (rounds 1-3's definition of a step: every step starts with an exposed admission on idle rows)."""
    wl.args.drain = True          # (the engine, its graphs and the scheduler's calibration are warm from the timed region)
    torch.cuda.synchronize(wl.dev)
    t_ = time.perf_counter()
    wl.steps_continuous(n_dr)
    torch.cuda.synchronize(wl.dev)
    d_ = (time.perf_counter() - t_) / n_dr
    wl.args.drain = False
    return {"workload": f"rounds 1-3's definition of a step, {n_dr} of them back to back: every step starts on an idle engine (its first admission is exposed) "
                        "and its last rows decode with nothing staged under them",
            "steps": n_dr, "tiles_per_s": round(wl.n_req / d_, 3), "ms_per_step": round(d_ * 1e3, 2)}


def more_rows():
    """Beyond the headline's 32 rows (NOT the headline: BASELINE.json configs[2] says batch = 32): the same workload through 64 and 128 batch rows per
    GPU (the reference's request-level mode keeps up to 128 requests in flight per worker, generate_scheduler.py:57).  Each point is this script
    run as a child process (own engine, 2 steps of 2 x rows requests, no side measurements)."""
    out = {}
    for rows_ in (64, 128):
        cmd = [PYTHON, os.path.join(ROOT, "bench.py"), "--batch", str(rows_), "--steps", "2", "--warmup", "1", "--waves", "2", "--no-latency",
               "--no-cpu-baseline", "--no-pmc"]
        try:
            j_ = last_json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=child_env()).stdout)
            ph = j_["phase_ms_per_step"]
            out[str(rows_)] = {"tiles_per_s": j_["value"], "ms_per_step": j_["ms_per_step"], "tiles_per_step": j_["config"]["tiles_per_gpu_per_step"],
                               "decode_step_ms_alone": j_["roofline"]["decode_step_ms"], "decode_step_ms_shared": ph["scheduler"]["decode_step_ms_shared"],
                               "gemv_replay_launch_us": j_["roofline"]["avg_launch_us_replay"], "decode_step_frac": j_["roofline"].get("decode_step_frac"),
                               "forward_mfma_frac": ph["forward_mfma_frac"], "vit_mfma_frac": ph["vit_mfma_frac"], "workspace_GB": j_["workspace_GB"]}
        except Exception as e_:  # noqa: BLE001
            out[str(rows_)] = {"error": f"{type(e_).__name__}: {e_}"[:300]}
    return out


def ragged(wl):
    """Admit-on-finish TIMED: the same requests with ragged answer lengths (per-request max_new uniform in [64, 192], mean 128, seeded), through the
    same scheduler, against static batches of B that each run to their longest answer.  The headline's rows all stop on the same step (EOS is
    ignored by the metric), so only this phase shows what refilling rows as they free up is worth."""
    from socioreasoner_amd.serving import ContinuousBatcher, Request
    a, B, n_req, eng = wl.args, wl.B, wl.n_req, wl.eng
    lens = np.random.default_rng(4000).integers(RAGGED_LO, RAGGED_HI + 1, n_req).tolist()
    poll = a.poll_ragged or a.poll

    def run(ov):
        cb = ContinuousBatcher(eng, eos=[], pad_id=0, steps_per_poll=poll, overlap=ov, admit_cus_per_se=wl.admit_share())
        reqs = [Request(ids=wl.ids[k], pos3=wl.pos3[k], max_new=int(lens[k]), images=wl.imgs[k], grids=[wl.grid] * wl.n_img) for k in range(n_req)]
        torch.cuda.synchronize(wl.dev)
        t_ = time.perf_counter()
        toks_ = cb.run(reqs)
        torch.cuda.synchronize(wl.dev)
        dt_ = time.perf_counter() - t_
        assert [len(t) for t in toks_] == lens
        return dt_, cb.stats["steps"]

    run(wl.overlap)                                   # warm (graphs, calibration)
    r_dt, r_steps = run(wl.overlap)
    torch.cuda.synchronize(wl.dev)                    # static batches: B requests at a time, every batch decodes until its longest answer is done
    t_ = time.perf_counter()
    s_steps = 0
    for lo in range(0, n_req, B):
        pix = torch.cat([eng.patchify(im) for grp in wl.imgs[lo:lo + B] for im in grp], dim=0)
        emb = eng.vit_forward(pix, [wl.grid] * (min(B, n_req - lo) * wl.n_img))
        eng.prefill(wl.ids[lo:lo + B], wl.pos3[lo:lo + B], emb)
        eng.decode(max(lens[lo:lo + B]))
        s_steps += max(lens[lo:lo + B])
    torch.cuda.synchronize(wl.dev)
    s_dt = time.perf_counter() - t_
    return {"workload": f"{n_req} requests, max_new uniform in [{RAGGED_LO}, {RAGGED_HI}] (mean {sum(lens) / len(lens):.1f}, seed 4000), {B} rows, "
                        f"{poll} decode steps per scheduling round",
            "continuous_tiles_per_s": round(n_req / r_dt, 3), "continuous_tokens_per_s": round(sum(lens) / r_dt, 1), "continuous_decode_steps": r_steps,
            "static_batches_tiles_per_s": round(n_req / s_dt, 3), "static_batches_decode_steps": s_steps, "gain": round(s_dt / r_dt, 4)}


def sam2(dev):
    """SAM2 (Hiera-L) behind seg_infer: the mask half of a tile in the reference's pipeline (seg_strategy.py:47-60) -- 756 x 756 image -> set_image,
    then decode + arg-max + resize + OR per object.  Timed beside the LM path (the metric's tile uses synthetic masks: SURVEY 8(D))."""
    from socioreasoner_amd import sam2 as _sam2
    from socioreasoner_amd import synthetic
    sg = _sam2.Sam2Geometry()
    simg = torch.from_numpy(synthetic.tile_pixels(0, 756, 756)).to(dev)
    simgs = [torch.from_numpy(synthetic.tile_pixels(i, 756, 756)).to(dev) for i in range(8)]
    sobj = [dict(point_coords=[[300 + 20 * k, 320]], point_labels=[1], box=[100 + 30 * k, 120, 420 + 30 * k, 600]) for k in range(4)]
    ssd = _sam2.synthetic_state_dict(sg)

    def mode(dtype):
        se = _sam2.Sam2Engine(sg, str(dev), dtype=dtype)
        se.load_state_dict(ssd)
        sacc = torch.zeros(756, 756, dtype=torch.uint8, device=dev)

        def tile():                      # the reference's loop: one image, one object at a time
            se.set_image(simg)
            for o in sobj:
                se.predict_or(sacc, **o)

        def tiles8():                    # 8 tiles per encoder pass, the 4 objects of a tile per decoder pass
            se.set_images(simgs)
            for b_ in range(8):
                se.select(b_)
                se.predict_or_many(sacc, sobj)
        tile()
        tiles8()
        t_ = {}
        for nm, fn, reps in (("set_image_ms", lambda: se.set_image(simg), 5), ("set_images_8_ms", lambda: se.set_images(simgs), 3),
                             ("predict_ms_per_object", lambda: se.predict_or(sacc, **sobj[0]), 20),
                             ("predict_ms_4_objects_one_pass", lambda: se.predict_or_many(sacc, sobj), 20),
                             ("tile_ms_4_objects", tile, 5), ("tiles8_ms_4_objects", tiles8, 3)):
            torch.cuda.synchronize(dev)
            t0_ = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(dev)
            t_[nm] = round((time.perf_counter() - t0_) / reps * 1e3, 3)
        del se
        torch.cuda.empty_cache()
        return dict(t_, tiles_per_s_4_objects=round(1e3 / t_["tile_ms_4_objects"], 2), tiles_per_s_4_objects_batched=round(8e3 / t_["tiles8_ms_4_objects"], 2))
    # Hiera-L encoder: 1.57 TFLOP of Linear layers + 0.21 TFLOP of attention per 1024 x 1024 input
    f32 = mode(torch.float32)
    f32["dtype"] = ("float32 (the reference's precision: seg_infer's default); Linear layers on the bf16 matrix pipe from an exact three-term bf16 split of "
                    "both operands (six partial products, float32 accumulation: csrc/sam_f32.hip k_gemm_f32s); attention on v_mfma_f32_16x16x4_f32")
    f32["encoder_TFLOPs_batched"] = round(1.78e12 * 8 / (f32["set_images_8_ms"] * 1e-3) / 1e12, 1)
    f32["encoder_vs_f32_mfma_peak_157TF"] = round(1.78e12 * 8 / (f32["set_images_8_ms"] * 1e-3) / 157.3e12, 4)
    b16 = mode(torch.bfloat16)
    b16["dtype"] = "bf16 storage / float32 accumulation (opt-in: sam2_compute_dtype bf16; masks differ from float32's inside the bf16 noise band)"
    return {"workload": "SAM2 Hiera-L (216.9 M parameters, random init), 756 x 756 tiles -> 1024 x 1024 input, box + click prompts, 3 masks + scores per "
                        "object, "
                        "best mask resized to 756 x 756 and OR-ed on the device; tile_ms = one tile and one object at a time (the reference's loop), "
                        "tiles8_ms = 8 tiles per encoder pass and a tile's 4 objects per decoder pass (what seg_infer runs)",
            "float32": f32, "bf16": b16}


def pipeline(n_samples, batch_order_too=False):
    """The reference's own two-stage pipeline (examples/infer/rlvr_megatron.yaml through SocioSegInferPipeline: two generate calls on a (map,
    satellite) pair each, two segment calls, four PNGs + two text files per sample) at the reference's scale and sampling parameters
    (rollout_batch_size 250, temperature 1 / top_p 0.8 / top_k 100: rlvr_megatron.yaml:89-95) with SAM2 DOING WORK: random weights emit no
    <answer>, so tools/run_example_small.py scripts the decoded answers (4 objects per stage; the LM still generates its 128 tokens per stage on
    the engine).  A child process (own engines); phase wall times from SocioSegInferPipeline.timing."""
    env = child_env(SCRIPTED_OBJECTS="4", SOCIOSEG_NUM_SAMPLES=str(n_samples), NEW_TOKENS=str(N_NEW), OUT="/tmp/sr_bench_pipeline_out",
                    ROLLOUT_BATCH=str(n_samples))
    script = [PYTHON, os.path.join(ROOT, "tools", "run_example_small.py")]
    try:
        r_ = subprocess.run(script, capture_output=True, text=True, timeout=900, env=env)
        p = last_json_line(r_.stdout)
        try:      # the same run with 128 LM batch rows instead of the example YAML's 32 (the reference's scheduler keeps up to 128 requests in flight)
            q = last_json_line(subprocess.run(script, capture_output=True, text=True, timeout=900, env=dict(env, MAX_BATCH="128")).stdout)
            p["with_128_lm_batch_rows"] = {k: q[k] for k in ("samples_per_s", "run_s", "wall_s_by_phase")}
        except Exception as e_:  # noqa: BLE001
            p["with_128_lm_batch_rows"] = {"error": f"{type(e_).__name__}: {e_}"[:200]}
        # --pipeline-ab: the reference's order of the host flow (two generate calls per batch, the host work between them) instead of the streamed
        # one.  Not in the default run, which has to stay within minutes: profiles/r06_pipeline250_ab.txt holds the A/B (tools/gpu_lease.sh pipeline250)
        if batch_order_too:
            try:
                q = last_json_line(subprocess.run(script, capture_output=True, text=True, timeout=900, env=dict(env, SOCIOSEG_STREAM="0")).stdout)
                p["batch_order_SOCIOSEG_STREAM_0"] = {k: q[k] for k in ("samples_per_s", "run_s", "wall_s_by_phase")}
            except Exception as e_:  # noqa: BLE001
                p["batch_order_SOCIOSEG_STREAM_0"] = {"error": f"{type(e_).__name__}: {e_}"[:200]}
        p["workload"] = (f"SocioSegInferPipeline.run() on {n_samples} synthetic SocioSeg samples, the shipped YAML (3B LM + SAM2 Hiera-L float32, synthetic "
                         "weights; rollout_batch_size = the sample count as in the reference YAML; "
                         "the YAML's sampling parameters), 128 new tokens per stage, decoded answers scripted to 4 objects per stage so that seg_infer encodes "
                         "every satellite image and decodes 4 prompts per stage and sample; streamed mode (one open request stream for both "
                         "stages, the host flow "
                         "under generation) unless the entry says otherwise")
        return p
    except Exception as e_:  # noqa: BLE001
        return {"error": f"{type(e_).__name__}: {e_}"[:300]}


def latency_b1(wl):
    """configs[1] beside the headline: one tile at a time on the same engine (batch-1 kernels, hipGraph decode)."""
    from socioreasoner_amd import lib as L
    zero = lambda: {"vit": 0.0, "prefill": 0.0, "decode": 0.0, "raster": 0.0}      # noqa: E731
    lat = zero()
    wl.step_static(1)
    torch.cuda.synchronize(wl.dev)
    t1 = time.perf_counter()
    k = max(wl.args.steps, 3)
    for _ in range(k):
        wl.step_static(1, lat)
    torch.cuda.synchronize(wl.dev)
    d1 = time.perf_counter() - t1
    # opt-in split-K of the small-M residual GEMMs (engine.hip prefill_splitk: off by default because it gives up bit-exact batch invariance):
    # one extra batch-1 step with it switched on, reported beside the default numbers
    sk = zero()
    os.environ["SR_SPLITK"] = "1"
    L.reload_switches()            # the library reads its switches once, not per call
    wl.step_static(1)
    wl.step_static(1, sk)
    os.environ.pop("SR_SPLITK")
    L.reload_switches()
    q = wl.quotes_mfma
    return {"workload": "BASELINE.json configs[1]: batch 1, one tile per step", "tiles_per_s": round(k / d1, 4), "ms_per_tile": round(d1 / k * 1e3, 3),
            "steps": k,
            "phase_ms": {n: round(v / k, 3) for n, v in lat.items()}, "decode_step_ms": round(lat["decode"] / k / (N_NEW - 1), 4),
            "vit_mfma_frac": frac_of_mfma_peak(VIT_GFLOP, lat["vit"] / k) if q else None,
            "prefill_mfma_frac": frac_of_mfma_peak(PREFILL_GFLOP, lat["prefill"] / k) if q else None,
            "opt_in_splitk": {"prefill_ms": round(sk["prefill"], 3), "prefill_mfma_frac": frac_of_mfma_peak(PREFILL_GFLOP, sk["prefill"]) if q else None,
                              "note": "SR_SPLITK=1: o_proj / down-projection of prefills <= 1024 rows split over K; not the default (float32 association "
                                      "differs "
                                      "from the batched kernels)"}}
