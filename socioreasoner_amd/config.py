"""Model geometry of the hot path (HF config fields of Qwen2.5-VL) and its mapping onto ``sr_config``."""
from __future__ import annotations

from dataclasses import dataclass, field


@dataclass
class VisionGeometry:
    depth: int = 32
    hidden_size: int = 1280
    num_heads: int = 16
    intermediate_size: int = 3420
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    window_size: int = 112
    fullatt_block_indexes: tuple = (7, 15, 23, 31)
    out_hidden_size: int = 2048
    in_channels: int = 3


@dataclass
class TextGeometry:
    num_hidden_layers: int = 36
    hidden_size: int = 2048
    num_attention_heads: int = 16
    num_key_value_heads: int = 2
    head_dim: int = 128
    intermediate_size: int = 11008
    vocab_size: int = 151936
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    mrope_section: tuple = (16, 24, 24)


@dataclass
class ModelGeometry:
    vision: VisionGeometry = field(default_factory=VisionGeometry)
    text: TextGeometry = field(default_factory=TextGeometry)
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653
    eos_token_id: int = 151645
    pad_token_id: int = 151643

    def param_specs(self):
        """(hf_name, shape, base) in the reference-era checkpoint naming
        (/root/reference/mcore_adapter/src/mcore_adapter/models/converter/template.py:845-899)."""
        v, t = self.vision, self.text
        out = []
        pd = v.in_channels * v.temporal_patch_size * v.patch_size * v.patch_size
        out.append(("visual.patch_embed.proj.weight", (v.hidden_size, pd), 0.0))
        for i in range(v.depth):
            p = f"visual.blocks.{i}."
            out += [(p + "norm1.weight", (v.hidden_size,), 1.0), (p + "norm2.weight", (v.hidden_size,), 1.0),
                    (p + "attn.qkv.weight", (3 * v.hidden_size, v.hidden_size), 0.0),
                    (p + "attn.qkv.bias", (3 * v.hidden_size,), 0.0),
                    (p + "attn.proj.weight", (v.hidden_size, v.hidden_size), 0.0),
                    (p + "attn.proj.bias", (v.hidden_size,), 0.0),
                    (p + "mlp.gate_proj.weight", (v.intermediate_size, v.hidden_size), 0.0),
                    (p + "mlp.gate_proj.bias", (v.intermediate_size,), 0.0),
                    (p + "mlp.up_proj.weight", (v.intermediate_size, v.hidden_size), 0.0),
                    (p + "mlp.up_proj.bias", (v.intermediate_size,), 0.0),
                    (p + "mlp.down_proj.weight", (v.hidden_size, v.intermediate_size), 0.0),
                    (p + "mlp.down_proj.bias", (v.hidden_size,), 0.0)]
        mh = v.hidden_size * v.spatial_merge_size ** 2
        out += [("visual.merger.ln_q.weight", (v.hidden_size,), 1.0), ("visual.merger.mlp.0.weight", (mh, mh), 0.0),
                ("visual.merger.mlp.0.bias", (mh,), 0.0), ("visual.merger.mlp.2.weight", (v.out_hidden_size, mh), 0.0),
                ("visual.merger.mlp.2.bias", (v.out_hidden_size,), 0.0),
                ("model.embed_tokens.weight", (t.vocab_size, t.hidden_size), 0.0)]
        qd, kd = t.num_attention_heads * t.head_dim, t.num_key_value_heads * t.head_dim
        for i in range(t.num_hidden_layers):
            p = f"model.layers.{i}."
            out += [(p + "input_layernorm.weight", (t.hidden_size,), 1.0),
                    (p + "self_attn.q_proj.weight", (qd, t.hidden_size), 0.0), (p + "self_attn.q_proj.bias", (qd,), 0.0),
                    (p + "self_attn.k_proj.weight", (kd, t.hidden_size), 0.0), (p + "self_attn.k_proj.bias", (kd,), 0.0),
                    (p + "self_attn.v_proj.weight", (kd, t.hidden_size), 0.0), (p + "self_attn.v_proj.bias", (kd,), 0.0),
                    (p + "self_attn.o_proj.weight", (t.hidden_size, qd), 0.0),
                    (p + "post_attention_layernorm.weight", (t.hidden_size,), 1.0),
                    (p + "mlp.gate_proj.weight", (t.intermediate_size, t.hidden_size), 0.0),
                    (p + "mlp.up_proj.weight", (t.intermediate_size, t.hidden_size), 0.0),
                    (p + "mlp.down_proj.weight", (t.hidden_size, t.intermediate_size), 0.0)]
        out.append(("model.norm.weight", (t.hidden_size,), 1.0))
        return out


def geometry_3b() -> ModelGeometry:
    """SocioReasoner-3B = Qwen2.5-VL-3B."""
    return ModelGeometry()


def geometry_tiny() -> ModelGeometry:
    """Small geometry with the true head dims (ViT 80, LM 128) used by the fast parity tests."""
    return ModelGeometry(
        vision=VisionGeometry(depth=4, hidden_size=320, num_heads=4, intermediate_size=220,
                              fullatt_block_indexes=(1, 3), out_hidden_size=512),
        text=TextGeometry(num_hidden_layers=3, hidden_size=512, num_attention_heads=4, num_key_value_heads=1,
                          intermediate_size=1000, vocab_size=2048),
        image_token_id=2040, video_token_id=2041, vision_start_token_id=2042, vision_end_token_id=2043,
        eos_token_id=2044, pad_token_id=2045)


def geometry_from_hf_config(cfg: dict) -> ModelGeometry:
    """``config.json`` of a Qwen2.5-VL checkpoint directory -> geometry.  Both layouts are read: the reference-era one (text fields at the
    top level, ``rope_scaling.mrope_section``; what ``vvangfaye/SocioReasoner-3B`` ships -- /root/reference/mcore_adapter/src/mcore_adapter/
    models/qwen2_5_vl/config_qwen2_5_vl.py:11-15 for the token ids) and the transformers-5 one (``text_config`` / ``rope_parameters``)."""
    vc = dict(cfg.get("vision_config") or {})
    tc = dict(cfg.get("text_config") or {})
    t = lambda k, d=None: tc.get(k, cfg.get(k, d))
    rope = t("rope_parameters") or t("rope_scaling") or {}
    dv, dt = VisionGeometry(), TextGeometry()
    heads = int(t("num_attention_heads", dt.num_attention_heads))
    hidden = int(t("hidden_size", dt.hidden_size))
    vision = VisionGeometry(depth=int(vc.get("depth", dv.depth)), hidden_size=int(vc.get("hidden_size", dv.hidden_size)),
                            num_heads=int(vc.get("num_heads", dv.num_heads)), intermediate_size=int(vc.get("intermediate_size", dv.intermediate_size)),
                            patch_size=int(vc.get("patch_size", dv.patch_size)), temporal_patch_size=int(vc.get("temporal_patch_size", dv.temporal_patch_size)),
                            spatial_merge_size=int(vc.get("spatial_merge_size", dv.spatial_merge_size)), window_size=int(vc.get("window_size", dv.window_size)),
                            fullatt_block_indexes=tuple(vc.get("fullatt_block_indexes", dv.fullatt_block_indexes)),
                            out_hidden_size=int(vc.get("out_hidden_size", hidden)), in_channels=int(vc.get("in_channels", vc.get("in_chans", dv.in_channels))))
    text = TextGeometry(num_hidden_layers=int(t("num_hidden_layers", dt.num_hidden_layers)), hidden_size=hidden, num_attention_heads=heads,
                        num_key_value_heads=int(t("num_key_value_heads", dt.num_key_value_heads)), head_dim=int(t("head_dim") or hidden // heads),
                        intermediate_size=int(t("intermediate_size", dt.intermediate_size)), vocab_size=int(t("vocab_size", dt.vocab_size)),
                        rms_norm_eps=float(t("rms_norm_eps", dt.rms_norm_eps)), rope_theta=float(rope.get("rope_theta") or t("rope_theta", dt.rope_theta)),
                        mrope_section=tuple(rope.get("mrope_section") or dt.mrope_section))
    d = ModelGeometry()
    eos = cfg.get("eos_token_id", tc.get("eos_token_id", d.eos_token_id))
    # HF ties the weights from the TOP-LEVEL flag (PreTrainedModel.tie_weights reads config.tie_word_embeddings); a re-saved config can carry a sub-config
    # default that differs, so the top level wins when it is there; checkpoints.py / load_safetensors_dir additionally refuse a checkpoint
    # whose index holds an lm_head.weight
    tie = cfg["tie_word_embeddings"] if "tie_word_embeddings" in cfg else tc.get("tie_word_embeddings", True)
    check_supported(ModelGeometry(vision=vision, text=text), tie_word_embeddings=tie)
    return ModelGeometry(vision=vision, text=text, image_token_id=int(cfg.get("image_token_id", d.image_token_id)),
                         video_token_id=int(cfg.get("video_token_id", d.video_token_id)), vision_start_token_id=int(cfg.get("vision_start_token_id", d.vision_start_token_id)),
                         vision_end_token_id=int(cfg.get("vision_end_token_id", d.vision_end_token_id)),
                         eos_token_id=int(eos[0] if isinstance(eos, (list, tuple)) else eos), pad_token_id=int(cfg.get("pad_token_id", tc.get("pad_token_id", d.pad_token_id)) or d.pad_token_id))


def check_supported(g: ModelGeometry, tie_word_embeddings=True) -> None:
    """Refuses, by name, every checkpoint geometry the engine's kernels cannot run CORRECTLY (a ``config.json`` of another
    Qwen2.5-VL size used to be accepted here and fail -- or, for an untied LM head, silently compute with the embedding matrix -- later).
    The limits are those of libsocior.so as built for SocioReasoner-3B (csrc/engine.hip ``validate``; include/socior.h)."""
    v, t = g.vision, g.text
    why = []
    if tie_word_embeddings is False:
        why.append("tie_word_embeddings is false: the engine uses model.embed_tokens.weight as the LM head (sr_load_weight ignores lm_head.weight), "
                   "an untied head would be replaced by the embedding matrix")
    if t.head_dim != 128:
        why.append(f"text head_dim {t.head_dim}: the attention and rotary kernels are built for 128")
    if t.hidden_size > 2048 or t.hidden_size % 64:
        why.append(f"text hidden_size {t.hidden_size}: the decode RMSNorm / fragment-ordered activation kernels take multiples of 64 up to 2048")
    if t.num_attention_heads % max(t.num_key_value_heads, 1) or t.num_attention_heads // max(t.num_key_value_heads, 1) > 16:
        why.append(f"GQA group {t.num_attention_heads} / {t.num_key_value_heads}: must divide, at most 16 query heads per KV head")
    if sum(t.mrope_section) != 64:
        why.append(f"mrope_section {tuple(t.mrope_section)} must sum to 64 (head_dim / 2)")
    if t.vocab_size % 16:
        why.append(f"vocab_size {t.vocab_size} must be a multiple of 16")
    if v.hidden_size % v.num_heads or v.hidden_size // v.num_heads not in (80, 128):
        why.append(f"vision head_dim {v.hidden_size}/{v.num_heads}: 80 or 128")
    if v.hidden_size % 64:
        why.append(f"vision hidden_size {v.hidden_size} must be a multiple of 64")
    if v.spatial_merge_size != 2:
        why.append(f"spatial_merge_size {v.spatial_merge_size}: 2")
    if v.out_hidden_size != t.hidden_size:
        why.append(f"vision out_hidden_size {v.out_hidden_size} != text hidden_size {t.hidden_size}")
    if len(v.fullatt_block_indexes) > 16:
        why.append("more than 16 full-attention blocks")
    if why:
        raise ValueError("this checkpoint geometry is not supported by the MI355X engine (built for SocioReasoner-3B = Qwen2.5-VL-3B):\n  - " + "\n  - ".join(why))


def geometry_to_hf_config(g: ModelGeometry) -> dict:
    """The reference-era ``config.json`` of this geometry (what a checkpoint directory carries)."""
    v, t = g.vision, g.text
    return {"architectures": ["Qwen2_5_VLForConditionalGeneration"], "model_type": "qwen2_5_vl", "hidden_size": t.hidden_size, "num_hidden_layers": t.num_hidden_layers,
            "num_attention_heads": t.num_attention_heads, "num_key_value_heads": t.num_key_value_heads, "intermediate_size": t.intermediate_size, "vocab_size": t.vocab_size,
            "rms_norm_eps": t.rms_norm_eps, "rope_theta": t.rope_theta, "rope_scaling": {"type": "mrope", "mrope_section": list(t.mrope_section)}, "tie_word_embeddings": True,
            "hidden_act": "silu", "torch_dtype": "bfloat16", "image_token_id": g.image_token_id, "video_token_id": g.video_token_id,
            "vision_start_token_id": g.vision_start_token_id, "vision_end_token_id": g.vision_end_token_id, "eos_token_id": g.eos_token_id, "pad_token_id": g.pad_token_id, "bos_token_id": g.pad_token_id,
            "vision_config": {"model_type": "qwen2_5_vl", "depth": v.depth, "hidden_size": v.hidden_size, "num_heads": v.num_heads, "intermediate_size": v.intermediate_size,
                              "patch_size": v.patch_size, "temporal_patch_size": v.temporal_patch_size, "spatial_merge_size": v.spatial_merge_size, "window_size": v.window_size,
                              "fullatt_block_indexes": list(v.fullatt_block_indexes), "out_hidden_size": v.out_hidden_size, "in_chans": v.in_channels, "hidden_act": "silu"}}
