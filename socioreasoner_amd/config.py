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
