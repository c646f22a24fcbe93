"""CLIP text encoder (and, for the SVD path, vision tower) on MI355X: drop-in for the `transformers` `CLIPTextModel` the reference loads next to the UNet
(/root/reference/train.py:88 `CLIPTextModel.from_pretrained(path, subfolder="text_encoder")`) and calls once per clip through
the pipeline's `_encode_prompt` (`text_encoder(ids)[0]`, diffusers TextToVideoSDPipeline).

Same constructor config fields (`CLIPTextConfig`), same state-dict keys (`text_model.embeddings.token_embedding.weight`,
`text_model.encoder.layers.N.self_attn.q_proj.weight`, ...), same call convention (`model(input_ids)[0]` /
`.last_hidden_state` = the final-layer-normed hidden states, `[1]` / `.pooler_output` = the row of the end-of-text token).
The arithmetic is the HIP token path: LayerNorm, ONE fused Q|K|V contraction with bias, the flash attention kernel with
causal masking (`AaAttention.causal`), output projection + residual in the contraction epilogue, fc1, the activation
(`gelu` or `quick_gelu`, `aa_blend`), fc2 + residual.  Unlike the UNets this component has its real reference in the build
container: the parity tests compare it with `transformers.CLIPTextModel` itself.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import AA_ACT_GELU, AA_ACT_QUICK_GELU
from .layers import LayerNorm, Linear, weights_key

_ACTS = {"gelu": AA_ACT_GELU, "quick_gelu": AA_ACT_QUICK_GELU}


class CLIPTextModelOutput(tuple):
    """(last_hidden_state, pooler_output) with attribute access, like transformers' BaseModelOutputWithPooling."""

    def __new__(cls, last_hidden_state, pooler_output):
        self = super().__new__(cls, (last_hidden_state, pooler_output))
        self.last_hidden_state, self.pooler_output = last_hidden_state, pooler_output
        return self


class CLIPAttention(nn.Module):
    def __init__(self, dim, heads, causal=True):
        super().__init__()
        if dim % heads or dim // heads not in (64, 80):
            raise ValueError("the MI355X attention kernels implement head_dim 64 (CLIP ViT-L/14, OpenCLIP ViT-H/14 text towers, "
                             "ViT-L/14 vision tower) and 80 (ViT-H/14 vision tower)")
        self.heads, self.head_dim, self.causal = heads, dim // heads, causal
        # (registration order k, v, q, out as in transformers: LoRA files list their adapters in module-traversal order, lora.py)
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (Linear(dim, dim) for _ in range(4))
        self._fused = None
        self._fused_key = None

    def _apply(self, fn, *a, **k):
        self._fused = None
        return super()._apply(fn, *a, **k)

    def fused(self):
        ps = (self.q_proj, self.k_proj, self.v_proj)
        key = weights_key(*[t for p in ps for t in (p.weight, p.bias)])
        if self._fused is None or self._fused_key != key:
            self._fused = ops.pack_weight(torch.cat([p.weight.detach() for p in ps]), torch.cat([p.bias.detach() for p in ps]))
            self._fused_key = key
        return self._fused

    def tokens(self, normed, residual, batch, length):
        d = self.q_proj.in_features
        qkv = ops.conv_gemm(normed, self.fused(), ops.linear_geom(normed.shape[0]))
        a = ops.attention(qkv, 0, qkv, d, qkv, 2 * d, self.heads, batch, 1, length, length, (length, 0, 1), (length, 0, 1),
                          causal=self.causal, head_dim=self.head_dim)
        return self.out_proj.tokens(a, residual=residual)


class CLIPMLP(nn.Module):
    def __init__(self, dim, inner, act):
        super().__init__()
        self.fc1, self.fc2, self.act = Linear(dim, inner), Linear(inner, dim), _ACTS[act]

    def tokens(self, x, residual):
        h = self.fc1.tokens(x)
        return self.fc2.tokens(ops.blend(h, act=self.act, out=h), residual=residual)


class CLIPEncoderLayer(nn.Module):
    def __init__(self, cfg, causal=True):
        super().__init__()
        self.self_attn = CLIPAttention(cfg.hidden_size, cfg.num_attention_heads, causal)
        self.layer_norm1 = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.mlp = CLIPMLP(cfg.hidden_size, cfg.intermediate_size, cfg.hidden_act)
        self.layer_norm2 = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)

    def tokens(self, x, batch, length):
        x = self.self_attn.tokens(self.layer_norm1.tokens(x), x, batch, length)
        return self.mlp.tokens(self.layer_norm2.tokens(x), x)


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.position_embedding = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)


class _Encoder(nn.Module):
    def __init__(self, cfg, causal=True):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(cfg, causal) for _ in range(cfg.num_hidden_layers)])


class _TextTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg)
        self.encoder = _Encoder(cfg)
        self.final_layer_norm = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class CLIPTextModel(nn.Module):
    config_name = "config.json"

    def __init__(self, vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16,
                 max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5, eos_token_id=2, **_):
        super().__init__()
        if hidden_act not in _ACTS:
            raise ValueError(f"hidden_act {hidden_act!r} is not implemented (gelu, quick_gelu)")
        self.config = SimpleNamespace(vocab_size=vocab_size, hidden_size=hidden_size, intermediate_size=intermediate_size,
                                      num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                                      max_position_embeddings=max_position_embeddings, hidden_act=hidden_act,
                                      layer_norm_eps=layer_norm_eps, eos_token_id=eos_token_id)
        self.text_model = _TextTransformer(self.config)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def load_state_dict(self, state_dict, strict=True, **k):
        state_dict = {n: v for n, v in state_dict.items() if not n.endswith("position_ids")}    # buffer of older checkpoints
        if state_dict and not any(n.startswith("text_model.") for n in state_dict):            # transformers >= 5 writes the keys
            state_dict = {"text_model." + n: v for n, v in state_dict.items()}                 # without the `text_model.` level
        return super().load_state_dict(state_dict, strict=strict, **k)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, variant=None, **overrides):
        """transformers directory layout: config.json + model.safetensors / pytorch_model.bin."""
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, cls.config_name)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        cfg = dict(cfg.get("text_config") or {}, **{k: v for k, v in cfg.items() if k != "text_config"})
        cfg.update(overrides)
        model = cls(**cfg)
        from ._ckpt import load_state
        state = load_state(root, "model", variant, "pytorch_model")
        model.load_state_dict(state)
        return model.to(torch_dtype) if torch_dtype is not None else model

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(dict(vars(self.config), architectures=["CLIPTextModel"], model_type="clip_text_model"), f, indent=2)
        from safetensors.torch import save_file
        save_file({k: v.contiguous().cpu() for k, v in self.state_dict().items()}, os.path.join(path, "model.safetensors"))

    def gradient_checkpointing_enable(self):        # inference-only implementation: accepted, no effect (train.py:110-114)
        return None

    def gradient_checkpointing_disable(self):
        return None

    def forward(self, input_ids, attention_mask=None, position_ids=None, **_):
        """input_ids [B, L] (L <= max_position_embeddings) -> (last_hidden_state [B, L, D], pooler_output [B, D]).
        `attention_mask` is accepted and, like the reference's call (no mask is passed, padding attends causally), unused."""
        if not input_ids.is_cuda and not _lib.host_pointers_ok():
            raise RuntimeError("animate_anything_amd.CLIPTextModel runs on the GPU only (no CPU fallback)")
        tm = self.text_model
        b, length = input_ids.shape
        pos = torch.arange(length, device=input_ids.device) if position_ids is None else position_ids.reshape(-1)[:length]
        x = (tm.embeddings.token_embedding.weight[input_ids] + tm.embeddings.position_embedding.weight[pos]).reshape(b * length, -1)
        x = x.contiguous()
        for layer in tm.encoder.layers:
            x = layer.tokens(x, b, length)
        x = tm.final_layer_norm.tokens(x).reshape(b, length, -1)
        eos = self.config.eos_token_id
        if eos == 2:                                  # (transformers: legacy configs mark the end of text as the largest token id)
            idx = input_ids.to(torch.int).argmax(dim=-1)
        else:
            idx = (input_ids.to(torch.int) == eos).int().argmax(dim=-1)
        return CLIPTextModelOutput(x, x[torch.arange(b, device=x.device), idx])


# ----------------------------------------------------------------------------------------- vision tower (SVD path)
class CLIPVisionModelOutput(SimpleNamespace):
    """transformers CLIPVisionModelOutput: `.image_embeds`, `.last_hidden_state`."""


class _VisionEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        from .layers import Conv2d
        self.class_embedding = nn.Parameter(torch.randn(cfg.hidden_size))
        self.patch_embedding = Conv2d(cfg.num_channels, cfg.hidden_size, cfg.patch_size, stride=cfg.patch_size, bias=False)
        n = (cfg.image_size // cfg.patch_size) ** 2
        self.position_embedding = nn.Embedding(n + 1, cfg.hidden_size)


class _VisionTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _VisionEmbeddings(cfg)
        self.pre_layrnorm = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)          # (sic: the transformers attribute name)
        self.encoder = _Encoder(cfg, causal=False)
        self.post_layernorm = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class CLIPVisionModelWithProjection(nn.Module):
    """Drop-in for the `transformers` `CLIPVisionModelWithProjection` that diffusers' StableVideoDiffusionPipeline carries as
    `image_encoder` (loaded with the pipeline at /root/reference/train_svd.py:85-91, called once per clip by `_encode_image`):
    `model(pixel_values).image_embeds`.  ViT-H/14 (1280 wide, 32 layers, 16 heads of 80 channels, gelu) and ViT-L/14 (1024 wide,
    heads of 64).  The 14x14 / stride-14 patch embedding is the implicit-GEMM contraction over the 3-channel image padded to 8
    channels; attention is not causal; the class token's final state goes through `post_layernorm` and `visual_projection`."""

    config_name = "config.json"

    def __init__(self, hidden_size=1280, intermediate_size=5120, projection_dim=1024, num_hidden_layers=32, num_attention_heads=16,
                 num_channels=3, image_size=224, patch_size=14, hidden_act="gelu", layer_norm_eps=1e-5, **_):
        super().__init__()
        if hidden_act not in _ACTS:
            raise ValueError(f"hidden_act {hidden_act!r} is not implemented (gelu, quick_gelu)")
        if num_channels > 8 or image_size % patch_size:
            raise ValueError("num_channels <= 8 and image_size % patch_size == 0 are implemented")
        self.config = SimpleNamespace(hidden_size=hidden_size, intermediate_size=intermediate_size, projection_dim=projection_dim,
                                      num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                                      num_channels=num_channels, image_size=image_size, patch_size=patch_size,
                                      hidden_act=hidden_act, layer_norm_eps=layer_norm_eps)
        self.vision_model = _VisionTransformer(self.config)
        self.visual_projection = Linear(hidden_size, projection_dim, bias=False)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def load_state_dict(self, state_dict, strict=True, **k):
        state_dict = {n: v for n, v in state_dict.items() if not n.endswith("position_ids")}
        return super().load_state_dict(state_dict, strict=strict, **k)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, variant=None, **overrides):
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, cls.config_name)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        cfg = dict(cfg.get("vision_config") or {}, **{k: v for k, v in cfg.items() if k != "vision_config"})
        cfg.update(overrides)
        model = cls(**cfg)
        from ._ckpt import load_state
        state = load_state(root, "model", variant, "pytorch_model")
        model.load_state_dict(state)
        return model.to(torch_dtype) if torch_dtype is not None else model

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(dict(vars(self.config), architectures=["CLIPVisionModelWithProjection"], model_type="clip_vision_model"), f, indent=2)
        from safetensors.torch import save_file
        save_file({k: v.contiguous().cpu() for k, v in self.state_dict().items()}, os.path.join(path, "model.safetensors"))

    def forward(self, pixel_values, **_):
        """pixel_values [B, 3, image_size, image_size] (CLIP-normalised) -> .image_embeds [B, projection_dim]."""
        if not pixel_values.is_cuda and not _lib.host_pointers_ok():
            raise RuntimeError("animate_anything_amd.CLIPVisionModelWithProjection runs on the GPU only (no CPU fallback)")
        cfg, vm = self.config, self.vision_model
        b, c, h, w = pixel_values.shape
        if h != cfg.image_size or w != cfg.image_size:
            raise ValueError(f"Input image size ({h}*{w}) doesn't match model ({cfg.image_size}*{cfg.image_size}).")
        dt = self.dtype
        x8 = torch.zeros(b, h, w, 8, dtype=dt, device=pixel_values.device)
        x8[..., :c] = pixel_values.to(dt).permute(0, 2, 3, 1)
        side = cfg.image_size // cfg.patch_size
        geom = ops.Geom(b, h, w, side, side, cfg.patch_size, 0, 0)
        patches = vm.embeddings.patch_embedding.tokens(x8.reshape(-1, 8), geom).reshape(b, side * side, -1)
        cls_tok = vm.embeddings.class_embedding.to(dt).expand(b, 1, -1)
        length = side * side + 1
        x = (torch.cat([cls_tok, patches], dim=1) + vm.embeddings.position_embedding.weight[:length].to(dt)).reshape(b * length, -1)
        x = vm.pre_layrnorm.tokens(x.contiguous())
        for layer in vm.encoder.layers:
            x = layer.tokens(x, b, length)
        last = x.reshape(b, length, -1)
        pooled = vm.post_layernorm.tokens(last[:, 0].contiguous())
        return CLIPVisionModelOutput(image_embeds=self.visual_projection.tokens(pooled), last_hidden_state=last)
