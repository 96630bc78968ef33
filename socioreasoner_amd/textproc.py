"""Offline stand-ins for the two Hugging Face objects the reference's host code is written against -- the tokenizer and the
Qwen2-VL ``processor`` (reference use: roll/datasets/collator.py:444-564, rlvr_socioseg_vlm_pipeline_infer.py:61-144,
714-760).  Neither the tokenizer files nor the network exist here, so the SocioSeg pipeline is driven by these when no
checkpoint directory is given; with a real checkpoint the HF objects are used instead and nothing in this file runs.

* ``ByteTokenizer``: UTF-8 bytes are ids 0..255, the Qwen special tokens keep the ids of the model geometry.
  ``decode(encode(s)) == s`` for every string, so parsers and prompt builders see real text.
* ``SyntheticProcessor``: the subset of the HF processor contract the reference calls: ``apply_chat_template`` (Qwen
  chat markup), ``__call__(images=, text=)`` -> ``input_ids / attention_mask / image_grid_thw`` with each
  ``<|image_pad|>`` expanded to one token per merged patch (hf: processing_qwen2_5_vl), and ``image_processor`` with the
  attributes ``process_image`` reads.  It does not produce ``pixel_values``: the engine patchifies on the device (K1).
"""
from __future__ import annotations

import re
from typing import Dict, List, Sequence

import numpy as np
import torch

from socioreasoner_amd import hostops
from socioreasoner_amd.config import ModelGeometry

IMAGE_PLACEHOLDER = "<|vision_start|><|image_pad|><|vision_end|>"


class Features(dict):
    """dict with the two BatchFeature methods the reference calls (collator.py:466-472)."""

    def convert_to_tensors(self, tensor_type="pt"):
        for k, v in list(self.items()):
            if not isinstance(v, torch.Tensor):
                self[k] = torch.as_tensor(np.asarray(v))
        return self


class ByteTokenizer:
    padding_side = "left"

    def __init__(self, geom: ModelGeometry):
        g = geom
        self.vocab_size = g.text.vocab_size
        self.special: Dict[str, int] = {
            "<|endoftext|>": g.pad_token_id, "<|im_end|>": g.eos_token_id,
            "<|vision_start|>": g.vision_start_token_id, "<|vision_end|>": g.vision_end_token_id,
            "<|image_pad|>": g.image_token_id, "<|video_pad|>": g.video_token_id,
        }
        # <|im_start|> sits right below <|im_end|> in the Qwen vocabulary (151644 / 151645)
        self.special["<|im_start|>"] = g.eos_token_id - 1 if g.eos_token_id - 1 not in self.special.values() else g.pad_token_id + 1
        assert len(set(self.special.values())) == len(self.special) and min(self.special.values()) >= 256
        self.by_id = {v: k for k, v in self.special.items()}
        self.eos_token_id, self.pad_token_id = g.eos_token_id, g.pad_token_id
        self.eos_token, self.pad_token = "<|im_end|>", "<|endoftext|>"
        self.additional_special_tokens: List[str] = []
        self.additional_special_tokens_ids: List[int] = []
        self._split = re.compile("(" + "|".join(re.escape(t) for t in sorted(self.special, key=len, reverse=True)) + ")")

    def convert_tokens_to_ids(self, token: str) -> int:
        return self.special[token]

    def encode(self, text: str, add_special_tokens: bool = False) -> List[int]:
        out: List[int] = []
        for piece in self._split.split(text):
            if piece in self.special:
                out.append(self.special[piece])
            elif piece:
                out.extend(piece.encode("utf-8"))
        return out

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = False) -> str:
        parts: List[str] = []
        run = bytearray()
        for t in (int(x) for x in ids):
            if t < 256:
                run.append(t)
                continue
            if run:
                parts.append(run.decode("utf-8", errors="replace"))
                run = bytearray()
            if t in self.by_id:
                if not skip_special_tokens:
                    parts.append(self.by_id[t])
            else:
                parts.append(f"<|unused_{t}|>")   # an id no text maps to (random-weight models emit these): the name write_checkpoint_dir gives it
        if run:
            parts.append(run.decode("utf-8", errors="replace"))
        return "".join(parts)

    def batch_decode(self, ids, skip_special_tokens: bool = False) -> List[str]:
        rows = ids.tolist() if hasattr(ids, "tolist") else ids
        return [self.decode(r, skip_special_tokens) for r in rows]

    def pad(self, features: Dict[str, List[List[int]]], padding="max_length", max_length=None, pad_to_multiple_of=None,
            return_tensors="pt") -> Dict[str, torch.Tensor]:
        """tokenizer.pad as the reference uses it (collator.py:519-526): side = self.padding_side; a row longer than
        max_length is an error (HF would return a ragged batch that cannot become a tensor)."""
        rows = [list(map(int, r)) for r in features["input_ids"]]
        L = max(len(r) for r in rows)
        if padding == "max_length":
            assert max_length is not None
            if L > max_length:
                raise ValueError(f"prompt of {L} tokens exceeds max_length {max_length}")
            L = max_length
        if pad_to_multiple_of:
            L = (L + pad_to_multiple_of - 1) // pad_to_multiple_of * pad_to_multiple_of
        ids = torch.full((len(rows), L), self.pad_token_id, dtype=torch.long)
        mask = torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            if self.padding_side == "left":
                ids[i, L - len(r):] = torch.tensor(r)
                mask[i, L - len(r):] = 1
            else:
                ids[i, :len(r)] = torch.tensor(r)
                mask[i, :len(r)] = 1
        return {"input_ids": ids, "attention_mask": mask}


class _ImageProcessorSpec:
    """The attributes of HF's Qwen2VLImageProcessor that process_image reads (reference :126-144)."""
    image_processor_type = "Qwen2VLImageProcessor"

    def __init__(self, geom: ModelGeometry):
        from PIL import Image
        self.patch_size = geom.vision.patch_size
        self.merge_size = geom.vision.spatial_merge_size
        self.temporal_patch_size = geom.vision.temporal_patch_size
        self.min_pixels, self.max_pixels = 56 * 56, 28 * 28 * 1280
        self.resample = Image.BICUBIC


class SyntheticProcessor:
    def __init__(self, geom: ModelGeometry):
        self.geom = geom
        self.tokenizer = ByteTokenizer(geom)
        self.image_processor = _ImageProcessorSpec(geom)
        self.image_token = "<|image_pad|>"

    def apply_chat_template(self, messages, tokenize: bool = False, add_generation_prompt: bool = True) -> str:
        """Qwen2.5-VL chat markup for text + image content lists."""
        assert not tokenize
        out = ["<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n"]
        for m in messages:
            out.append(f"<|im_start|>{m['role']}\n")
            content = m["content"]
            if isinstance(content, str):
                out.append(content)
            else:
                for c in content:
                    out.append(IMAGE_PLACEHOLDER if c.get("type") == "image" else c.get("text", ""))
            out.append("<|im_end|>\n")
        if add_generation_prompt:
            out.append("<|im_start|>assistant\n")
        return "".join(out)

    def image_grid(self, image) -> tuple:
        ip = self.image_processor
        w, h = image.size
        rh, rw = hostops.smart_resize(h, w, factor=ip.patch_size * ip.merge_size, min_pixels=ip.min_pixels, max_pixels=ip.max_pixels)
        return (1, rh // ip.patch_size, rw // ip.patch_size)

    def __call__(self, images=None, text=None, **_) -> Features:
        if images is not None and not isinstance(images, (list, tuple)):
            images = [images]
        grids = [self.image_grid(im) for im in (images or [])]
        text = text if isinstance(text, str) else (text[0] if text else "")
        n_slots = text.count(self.image_token)
        if n_slots != len(grids):
            raise ValueError(f"text has {n_slots} image placeholders for {len(grids)} images")
        m2 = self.image_processor.merge_size ** 2
        pieces = text.split(self.image_token)
        full = pieces[0]
        for g, rest in zip(grids, pieces[1:]):
            full += self.image_token * (g[0] * g[1] * g[2] // m2) + rest
        ids = self.tokenizer.encode(full)
        f = Features(input_ids=[ids], attention_mask=[[1] * len(ids)])
        if grids:
            f["image_grid_thw"] = torch.tensor(grids, dtype=torch.long)
        return f


def load_hf_processor(path: str):
    """The checkpoint's own processor (reference: ``default_processor_provider`` -> ``AutoProcessor.from_pretrained``,
    /root/reference/roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py:518-521).  ``AutoProcessor`` first; where torchvision is absent
    (this image) transformers 5 cannot build Qwen2.5-VL's VIDEO processor and AutoProcessor raises ImportError -- then the same
    ``Qwen2_5_VLProcessor`` class is assembled from the checkpoint's files with HF's PIL image backend and no video processor (the
    pipeline never feeds videos): HF's own ``__call__`` / ``apply_chat_template`` / image processor run either way."""
    import os
    try:
        from transformers import AutoProcessor
        return AutoProcessor.from_pretrained(path)
    except ImportError:
        pass
    import json
    import types
    from transformers import AutoTokenizer
    from transformers.models.qwen2_5_vl.processing_qwen2_5_vl import Qwen2_5_VLProcessor
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil

    class _NoVideoProcessor(Qwen2_5_VLProcessor):
        def check_argument_for_proper_class(self, argument_name, argument):
            if argument_name == "video_processor":
                return type(argument)
            return super().check_argument_for_proper_class(argument_name, argument)
    tok = AutoTokenizer.from_pretrained(path)
    ip = Qwen2VLImageProcessorPil.from_pretrained(path)
    tmpl = getattr(tok, "chat_template", None)
    for name in ("chat_template.jinja", "chat_template.json"):
        f = os.path.join(path, name)
        if os.path.exists(f):
            txt = open(f, encoding="utf-8").read()
            tmpl = json.loads(txt)["chat_template"] if name.endswith(".json") else txt
            break
    video = types.SimpleNamespace(merge_size=getattr(ip, "merge_size", 2), temporal_patch_size=getattr(ip, "temporal_patch_size", 2))
    return _NoVideoProcessor(image_processor=ip, tokenizer=tok, video_processor=video, chat_template=tmpl)


QWEN_CHAT_TEMPLATE = (
    "{% for message in messages %}{% if loop.first and message['role'] != 'system' %}<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n{% endif %}"
    "<|im_start|>{{ message['role'] }}\n{% if message['content'] is string %}{{ message['content'] }}<|im_end|>\n{% else %}{% for content in message['content'] %}"
    "{% if content['type'] == 'image' or 'image' in content or 'image_url' in content %}<|vision_start|><|image_pad|><|vision_end|>"
    "{% elif 'text' in content %}{{ content['text'] }}{% endif %}{% endfor %}<|im_end|>\n{% endif %}{% endfor %}"
    "{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}")


def write_checkpoint_dir(path: str, geom: ModelGeometry, tensors: Dict[str, torch.Tensor]) -> None:
    """Writes an HF-style Qwen2.5-VL checkpoint DIRECTORY for ``geom`` around ``tensors`` (HF-named weights): model.safetensors, config.json,
    preprocessor_config.json, a ``tokenizers`` byte-level BPE tokenizer.json whose ids are ByteTokenizer's (bytes = 0..255, the Qwen special
    tokens at the geometry's ids, no merges) and the Qwen2-VL chat template.  Real SocioReasoner-3B files cannot be fetched offline; this
    is how the checkpoint branches of the pipeline (AutoProcessor / AutoTokenizer / safetensors loader / config.json geometry) are driven
    end to end by tests and tools -- the files have the real ones' names and schema."""
    import json
    import os
    from safetensors.torch import save_file
    from tokenizers import AddedToken, Tokenizer, decoders, models, pre_tokenizers
    from socioreasoner_amd.config import geometry_to_hf_config
    os.makedirs(path, exist_ok=True)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(path, "model.safetensors"))
    json.dump(geometry_to_hf_config(geom), open(os.path.join(path, "config.json"), "w"), indent=1)
    # GPT-2's byte <-> printable-character table (the alphabet of every byte-level BPE, Qwen's included)
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    ref = ByteTokenizer(geom)
    vocab = {chr(c): b for b, c in zip(bs, cs)}
    taken = set(ref.special.values())
    for i in range(256, geom.text.vocab_size):
        if i not in taken:
            vocab[f"<|unused_{i}|>"] = i
    vocab.update(ref.special)
    tok = Tokenizer(models.BPE(vocab=vocab, merges=[]))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    tok.decoder = decoders.ByteLevel()
    tok.add_special_tokens([AddedToken(k, special=True, normalized=False) for k in ref.special])
    tok.save(os.path.join(path, "tokenizer.json"))
    json.dump({"tokenizer_class": "Qwen2TokenizerFast", "eos_token": ref.eos_token, "pad_token": ref.pad_token, "model_max_length": 32768,
               "additional_special_tokens": [k for k in ref.special if k not in (ref.eos_token, ref.pad_token)], "chat_template": QWEN_CHAT_TEMPLATE},
              open(os.path.join(path, "tokenizer_config.json"), "w"), indent=1)
    v = geom.vision
    json.dump({"image_processor_type": "Qwen2VLImageProcessor", "processor_class": "Qwen2_5_VLProcessor", "patch_size": v.patch_size, "merge_size": v.spatial_merge_size,
               "temporal_patch_size": v.temporal_patch_size, "min_pixels": 3136, "max_pixels": 12845056, "image_mean": [0.48145466, 0.4578275, 0.40821073],
               "image_std": [0.26862954, 0.26130258, 0.27577711], "do_resize": True, "do_rescale": True, "do_normalize": True, "do_convert_rgb": True, "resample": 3,
               "rescale_factor": 1 / 255}, open(os.path.join(path, "preprocessor_config.json"), "w"), indent=1)
    open(os.path.join(path, "chat_template.jinja"), "w", encoding="utf-8").write(QWEN_CHAT_TEMPLATE)
