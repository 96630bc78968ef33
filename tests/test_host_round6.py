"""Round 6 CPU tests: the library's dynamic symbol table, the lost experiments are out of the product library, bench/ modules."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "socioreasoner_amd", "libsocior.so")


def _header_symbols():
    hdr = open(os.path.join(ROOT, "include", "socior.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)                      # (comments mention entry points of other rounds)
    return set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", hdr))


def test_library_exports_exactly_the_header():
    """A C ABI exports its header and nothing else (~70 mangled internals sat next to the sr_* symbols): the library is
    built with -fvisibility=hidden + a linker version script (csrc/exports.map), and `nm -D` must list exactly the functions include/socior.h declares."""
    out = subprocess.run(["nm", "-D", "--defined-only", SO], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    declared = _header_symbols()
    assert exported == declared, {"exported but not declared": sorted(exported - declared), "declared but not exported": sorted(declared - exported)}
    assert all(s.startswith("sr_") for s in exported)


def test_lost_experiments_are_not_in_the_product_library():
    """Round 5 re-grew what round 4 had removed: kernels and switches of experiments that were measured slower (in-launch RMSNorm heads / tails of the
    decode GEMVs, the counted form of the row-group GEMV, pre-split float32 weight planes).  They live as patches under tools/experiments/ now; neither
    the sources nor the built library may name them."""
    src = ""
    for d, _, files in os.walk(os.path.join(ROOT, "socioreasoner_amd")):
        for f in files:
            if f.endswith((".hip", ".h", ".py")):
                src += open(os.path.join(d, f), errors="replace").read()
    src += open(os.path.join(ROOT, "include", "socior.h")).read()
    for name in ("SR_TAIL_NORM", "SR_HEAD_NORM", "SR_SAM_PRESPLIT", "GemvTail", "GemvHead", "gemv_tail_rmsnorm", "sr_tail_timeouts"):
        assert name not in src, name
    blob = open(SO, "rb").read()
    for name in (b"SR_TAIL_NORM", b"SR_HEAD_NORM", b"gemv_tail_rmsnorm"):
        assert name not in blob, name
    # round 6's own lost experiment: the RMSNorm inside the x-stationary gate/up launch (k_gemv_px<.., NORM>, bits 3 / 4 of SR_GEMV_XLDS)
    for name in ("ln2_px", "rms_ss8", "bool NORM"):
        assert name not in src, name
    assert not any(b"k_gemv_pxILb" in blob and tag in blob for tag in (b"k_gemv_pxILb0ELi8ELb1", b"k_gemv_pxILb1ELi8ELb1")), "k_gemv_px<.., NORM> is in the library"
    for name in ("ctx_grid", "SR_ATTN_CTX_GRID"):
        assert name not in src, name
    for patch in ("gemv_tail_head_rmsnorm.patch", "gemv_px_norm_prologue.patch", "attn_decode_ctx_grid.patch"):
        assert os.path.exists(os.path.join(ROOT, "tools", "experiments", patch)), patch


# ------------------------------------------------------------------------------------------------ N2: the ground-truth box filter is cv2.contourArea, not the pixel count
def test_gt_boxes_follow_cv2_contour_area_on_hand_computed_cases():
    """Reference rlvr_socioseg_vlm_pipeline_infer.py:156-184 keeps a ground-truth blob when cv2.contourArea(outer contour) > 10 -- the polygon through
    the border pixels' CENTRES -- not when it has more than 10 pixels (rounds 1-5).  cv2 is not installed: the cases are computed by hand.
      filled w x h rectangle            -> (w - 1)(h - 1)            4 x 4: 9 (dropped, 16 pixels), 5 x 4: 12 (kept), 12 x 2: 11 (kept), 11 x 2: 10 (dropped)
      one-pixel line of any length      -> 0 (traced out and back)   40 pixels: dropped
      diamond |x| + |y| <= r            -> 2 r^2                     r = 2 (13 pixels): 8 dropped, r = 3 (25 pixels): 18 kept
      ring (a hole inside)              -> the OUTER polygon only    7 x 7 frame of width 1 (24 pixels): 36 kept
      a blob inside a ring's hole       -> no external contour (RETR_EXTERNAL): dropped whatever its size
      L of width 1 (two lines)          -> 0.5: on the way back the 8-connected trace cuts the corner pixel diagonally (one half-pixel triangle): dropped"""
    import json
    import numpy as np
    from socioreasoner_amd import socioseg_data as D
    rect = lambda w, h: np.ones((h, w), bool)      # noqa: E731
    for w, h, want in ((4, 4, 9.0), (5, 4, 12.0), (12, 2, 11.0), (11, 2, 10.0), (40, 1, 0.0), (1, 1, 0.0), (2, 1, 0.0)):
        assert D.contour_area(D.outer_border(rect(w, h), 0, 0)) == want, (w, h)

    def diamond(r):
        yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
        return (abs(yy) + abs(xx)) <= r
    for r in (1, 2, 3, 5):
        d = diamond(r)
        assert D.contour_area(D.outer_border(d, 0, r)) == 2.0 * r * r, r
    ell = np.zeros((9, 9), bool)
    ell[0:9, 0] = True
    ell[8, 0:9] = True
    assert D.contour_area(D.outer_border(ell, 0, 0)) == 0.5
    m = np.zeros((96, 96), np.uint8)
    m[2:6, 2:6] = 255             # 4 x 4: area 9 -> dropped (16 pixels: rounds 1-5 kept it)
    m[2:6, 10:15] = 255           # 5 x 4: area 12 -> kept
    m[10, 2:42] = 255             # 40-pixel line -> dropped (rounds 1-5 kept it)
    m[20:27, 20:27] = 255         # ring ...
    m[21:26, 21:26] = 0
    m[23, 23] = 255               # ... with a pixel in its hole
    m[40:60, 40:70] = 255         # big block with a big hole that holds a 6 x 6 blob: the blob has no external contour
    m[44:56, 44:66] = 0
    m[47:53, 50:56] = 255
    yy, xx = np.mgrid[0:96, 0:96]
    m[(abs(yy - 80) + abs(xx - 12)) <= 2] = 255       # diamond r = 2: 8 -> dropped
    m[(abs(yy - 80) + abs(xx - 30)) <= 3] = 255       # diamond r = 3: 18 -> kept
    from PIL import Image
    got = json.loads(D.get_bboxes([Image.fromarray(m, mode="L")])[0])
    # reverse raster order of each kept component's first pixel (OpenCV links new contours in front)
    assert got == [{"bbox_2d": [27, 77, 34, 84]}, {"bbox_2d": [40, 40, 70, 60]}, {"bbox_2d": [20, 20, 27, 27]}, {"bbox_2d": [10, 2, 15, 6]}], got
    assert D.count_components([Image.fromarray(m, mode="L")]) == [9]          # connectedComponentsWithStats counts every blob, holes' contents included


def test_rope_index_1d_equals_get_rope_index_on_random_prompt_structures():
    """hostops.rope_index_1d (numpy, one unpadded sequence: what Mi355xStrategy._prepare calls per request) against hostops.get_rope_index (the torch form that is
    pinned to the reference's own function by tests/golden/index.npz): text-only, one / two / three images of different grids, images back to back, an image
    token run that is NOT preceded by <vision_start> (plain text to both), text after the last image."""
    import numpy as np
    import torch
    from socioreasoner_amd import hostops
    IMG, VS = 151655, 151652
    rng = np.random.default_rng(7)
    for case in range(60):
        n_img = int(rng.integers(0, 4))
        grids, ids = [], []
        for k in range(n_img):
            ids += rng.integers(0, 1000, int(rng.integers(0 if k else 1, 12))).tolist()
            t, h, w = 1, 2 * int(rng.integers(1, 9)), 2 * int(rng.integers(1, 9))
            grids.append((t, h, w))
            ids += [VS] + [IMG] * (t * h * w // 4) + [151653]
        ids += rng.integers(0, 1000, int(rng.integers(0, 20))).tolist()
        if case % 7 == 3:
            ids += [5, IMG, IMG, 6]                      # a stray run without <vision_start>: the rule treats it as text
        ids = np.asarray(ids or [1], dtype=np.int64)
        want, _ = hostops.get_rope_index(torch.from_numpy(ids)[None], grids or None, None, image_token_id=IMG, vision_start_token_id=VS)
        got = hostops.rope_index_1d(ids, grids or None, image_token_id=IMG, vision_start_token_id=VS)
        assert got.dtype == np.int64 and np.array_equal(got, want[:, 0].numpy()), case


# ------------------------------------------------------------------------------------------------ the scheduler's CU hint (sr_rows_set_cus)
def test_scheduler_tells_the_engine_the_cu_count_only_when_it_changes():
    """ContinuousBatcher._set_cus / _hand_back (socioreasoner_amd/serving.py): the engine hears about the decode stream's CU count when decode moves between the unmasked
    and the masked stream -- not per chunk -- and is told "whole chip" again when the scheduler hands the stream back (the engine then replays the whole-chip form of
    its decode step for whoever calls next)."""
    from socioreasoner_amd.serving import ContinuousBatcher

    class Eng:
        def __init__(self):
            self.calls = []

        def rows_set_cus(self, n):
            self.calls.append(n)

    cb = ContinuousBatcher.__new__(ContinuousBatcher)
    cb.engine = Eng()
    for n in (0, 0, 160, 160, 160, 0, 176, 176):
        cb._set_cus(n)
    assert cb.engine.calls == [160, 0, 176]
    assert cb.engine._decode_cus == 176
    # hand-back: only an overlapped scheduler that decoded on its own streams has anything to undo
    cb.overlap, cb._dec_last = False, None
    cb._hand_back()
    assert cb.engine.calls == [160, 0, 176]
    # the decode CU count of a split = the complement of the admission mask
    from socioreasoner_amd.streams import split_masks
    for share, want in ((3, 160), (2, 192), (2.5, 176), (4, 128)):
        _, dec = split_masks(8, share)
        assert sum(bin(w).count("1") for w in dec) == want


def test_scheduler_takes_no_cost_measurement_next_to_foreign_gpu_work():
    """serving.foreign_gpu_load brackets GPU work the scheduler does not own (SAM2's prefetched encoder): no measurement starts inside the bracket, one in flight
    when the bracket opens or closes is dropped, and afterwards measurements are taken again."""
    from socioreasoner_amd import serving
    cb = serving.ContinuousBatcher.__new__(serving.ContinuousBatcher)
    cb._meas_epoch = {}
    assert cb._meas_begin("step") and cb._meas_clean("step")
    assert cb._meas_begin("adm")
    serving.foreign_gpu_load(True)
    try:
        assert not cb._meas_clean("adm")            # foreign work started under it
        assert not cb._meas_begin("dec_sh")         # none starts next to it
    finally:
        serving.foreign_gpu_load(False)
    assert cb._meas_begin("dec_sh") and cb._meas_clean("dec_sh")
    assert cb._meas_begin("adm_sh")
    serving.foreign_gpu_load(True)
    serving.foreign_gpu_load(False)
    assert not cb._meas_clean("adm_sh")             # ... or started AND ended under it
    assert serving._FOREIGN["active"] == 0


def test_request_stream_serves_two_waves_of_prompts_on_one_open_server(tmp_path):
    """GenerateScheduler.open_stream (the streamed two-stage pipeline's generation interface): ONE start_server / stop_server pair, prompts added under
    caller-chosen ids in two waves -- the second while answers of the first are being collected -- and every answer equals what a level-0 generate call
    returns for that prompt.  The request loop takes a burst of queued commands in one round (an ADD burst followed by STOP still answers every ADD)."""
    import numpy as np
    import torch
    from roll.distributed.scheduler.generate_scheduler import GenerateScheduler, assemble_responses
    from roll.distributed.scheduler.protocol import DataProto
    from roll.distributed.strategy.mi355x_strategy import Mi355xStrategy
    from roll.pipeline.base_worker import ActorWorker
    from socioreasoner_amd import hostops
    from socioreasoner_amd.config import geometry_tiny
    from socioreasoner_amd.textproc import ByteTokenizer
    from tests.test_host_cpu import _tiny_cfg
    cfg = _tiny_cfg(tmp_path, prompt_length=12, response_length=6)
    tok = ByteTokenizer(geometry_tiny())
    servers = []

    class Fake(Mi355xStrategy):          # the real request loop (start_server / add_request) over a scripted generate
        max_batch = 2

        def initialize(self, model_provider=None):
            import queue
            self.command_queue, self.tokenizer = queue.Queue(), tok

        def start_server(self, data, request_complete_callback):
            servers.append(1)
            return super().start_server(data, request_complete_callback)

        def generate(self, batch, generation_config):
            ids, mask = batch.batch["input_ids"], batch.batch["attention_mask"]
            rows = []
            for r, m in zip(ids, mask):
                p = r[m.bool()].tolist()
                rows.append(list(reversed(p))[: 1 + len(p) % 4] + [tok.eos_token_id])
            out = hostops.gather_outputs_to_pad_tensor(rows, generation_config["pad_token_id"], device=ids.device)
            return hostops.concatenate_input_and_output(ids, out, 1)

    assert Fake.request_stream
    w = ActorWorker(cfg.actor_infer, cfg, 0, 1, 0, "actor_infer")
    w.strategy = Fake(w)
    w.strategy.initialize()
    w.tokenizer = tok
    rng = np.random.default_rng(1)

    def prompts(lengths):
        ids = torch.full((len(lengths), 12), tok.pad_token_id, dtype=torch.long)
        mask = torch.zeros(len(lengths), 12, dtype=torch.long)
        for i, n in enumerate(lengths):
            ids[i, 12 - n:] = torch.from_numpy(rng.integers(0, 250, n))
            mask[i, 12 - n:] = 1
        pos = (mask.cumsum(-1) - 1).clamp(min=0)[:, None, :].repeat(1, 3, 1)
        return DataProto(batch={"input_ids": ids, "attention_mask": mask, "position_ids": pos}, non_tensor_batch={})
    a, b = prompts([12, 7, 9, 3, 10]), prompts([5, 11, 2])
    sched = GenerateScheduler()
    want_a = sched.generate(DataProto(batch={k: v.clone() for k, v in a.batch.items()}, non_tensor_batch={}), w, cfg)
    want_b = sched.generate(DataProto(batch={k: v.clone() for k, v in b.batch.items()}, non_tensor_batch={}), w, cfg)
    st = sched.open_stream(w, cfg)
    st.add(list(range(5)), a)
    got = dict(st.collect())
    st.add([100, 101, 102], b)                         # second wave, same server
    while len(got) < 8:
        got.update(dict(st.collect()))
    st.close()
    assert servers == [1] and st.in_flight == 0 and sorted(got) == [0, 1, 2, 3, 4, 100, 101, 102]
    out_a = assemble_responses(a, [got[i] for i in range(5)], w, cfg)
    out_b = assemble_responses(b, [got[100 + i] for i in range(3)], w, cfg)
    for k in want_a.batch:
        if k != "prompt_id":
            assert torch.equal(out_a.batch[k], want_a.batch[k]) and torch.equal(out_b.batch[k], want_b.batch[k]), k
    # a burst: three ADDs and the STOP queued before the loop has started -- every ADD is answered before the loop ends
    from roll.utils.functionals import GenerateRequestType
    answers = []
    for i in range(3):
        w.strategy.add_request(GenerateRequestType.ADD, DataProto(batch={k: v[i:i + 1] for k, v in b.batch.items()}, non_tensor_batch={},
                                                                  meta_info={"request_id": i, "generation_config": dict(st.gc, eos_token_id=[tok.eos_token_id], pad_token_id=tok.pad_token_id)}))
    w.strategy.add_request(GenerateRequestType.STOP, None)
    w.strategy.start_server(DataProto(meta_info={}), lambda data: answers.append((data.meta_info["request_id"], data.meta_info["output_token_ids"][0])))
    assert dict(answers) == {i: got[100 + i] for i in range(3)}


def test_collating_a_batch_in_pieces_gives_the_rows_of_the_whole_batch():
    """The streamed pipeline collates a rollout batch in pieces (the first piece's prompts are prefilled while the rest is collated): with padding to
    max_length every row is padded and indexed on its own, so the concatenated pieces must BE the whole batch's collation -- ragged image sizes included."""
    import numpy as np
    import torch
    from roll.datasets.collator import DataCollatorWithPaddingForMultiSeg
    from roll.distributed.scheduler.protocol import DataProto
    from roll.pipeline.rlvr import rlvr_socioseg_vlm_pipeline_infer as P
    from socioreasoner_amd import socioseg_data
    from socioreasoner_amd.config import geometry_tiny
    from socioreasoner_amd.textproc import SyntheticProcessor
    proc = SyntheticProcessor(geometry_tiny())
    proc.image_processor.max_pixels, proc.image_processor.min_pixels = 768 * 768, 56 * 56
    samples = socioseg_data.synthetic_socioseg(5, size=112)
    samples[1]["sat_image"] = samples[1]["sat_image"].resize((150, 100))
    samples[3]["map_image"] = samples[3]["map_image"].resize((84, 140))
    enc = P.encode_function({k: [s[k] for s in samples] for k in samples[0]}, proc)
    coll = DataCollatorWithPaddingForMultiSeg(tokenizer=proc.tokenizer, processor=proc, extra_data_provider=P.get_extra_data_provider(processor=proc),
                                              max_length=900, image_key="image", padding="max_length", gt_object_key="gt_object", gt_bbox_key="gt_bbox")
    rows = [{k: v[i] for k, v in enc.items()} for i in range(5)]
    whole = DataProto.from_single_dict(coll(rows))
    pieces = DataProto.concat([DataProto.from_single_dict(coll(rows[:3])), DataProto.from_single_dict(coll(rows[3:]))])
    assert set(whole.batch) == set(pieces.batch) and set(whole.non_tensor_batch) == set(pieces.non_tensor_batch)
    for k in whole.batch:
        assert torch.equal(whole.batch[k], pieces.batch[k]), k
    for k, v in whole.non_tensor_batch.items():
        w = pieces.non_tensor_batch[k]
        assert w.dtype == object and w.shape == v.shape == (5,), k
    for i in range(5):
        assert whole.non_tensor_batch["id"][i] == pieces.non_tensor_batch["id"][i]
        assert whole.non_tensor_batch["multi_modal_map_data"][i]["prompt_token_ids"] == pieces.non_tensor_batch["multi_modal_map_data"][i]["prompt_token_ids"]
        assert torch.equal(whole.non_tensor_batch["multi_modal_map_inputs"][i]["image_grid_thw"], pieces.non_tensor_batch["multi_modal_map_inputs"][i]["image_grid_thw"])
        assert whole.non_tensor_batch["seg_image"][i] is pieces.non_tensor_batch["seg_image"][i]      # the dataset's own objects (the 756-resize memo is keyed by them)
