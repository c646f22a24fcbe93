"""LoRA for inference by FOLDING the adapters into the base weights at load time: drop-in for the inference half of the
reference's cloneofsimo-style LoRA (/root/reference/utils/lora.py: `inject_inferable_lora` :482-526,
`monkeypatch_or_replace_lora_extended` :861-977, `collapse_lora` :780-815; caller /root/reference/train_lora.py:909-917).

The reference replaces every targeted nn.Linear / nn.Conv2d / nn.Conv3d by a `LoraInjected*` wrapper computing
    base(x) + scale * lora_up(lora_down(x))          (dropout is the identity in eval, the selector is the identity)
which is linear in x, so W' = W + scale * (up.flatten(1) @ down.flatten(1)).reshape(W.shape) gives the same function with
NO extra launches: the HIP kernels keep seeing one packed weight per layer (`collapse_lora` does the same fold, :794-815).

File format (`save_lora_weight`, utils/lora.py:569-581): a `.pt` list [up_0, down_0, up_1, down_1, ...] (fp32) in the order
`_find_modules` visits the injected layers: for every module whose CLASS NAME is in `target_replace_module` (in
`model.modules()` order), its `named_modules()` in order.  Which layers carry an adapter depends on the classes diffusers
used when the file was written: the reference injects only layers whose class is EXACTLY nn.Linear / nn.Conv2d / nn.Conv3d
(utils/lora.py:410,422,441,880-937; the `else: continue` branch "for pretrained models based on zeroscope_v2_576w" skips
diffusers' LoRACompatibleLinear / LoRACompatibleConv subclasses).  With diffusers==0.24.0 (requirements.txt:4) the plain
layers of the UNet are [D-0.24, from memory]: conv_in, conv_in2, conv_out, time_embedding.cond_proj, motion_embedding.{0,2},
every TemporalConvLayer Conv3d and the proj_in / proj_out of every TransformerTemporalModel.  Both layouts are accepted and
told apart by the number of adapter pairs in the file ("plain" = the list above, "all" = every Linear / Conv layer); every
pair is shape-checked against the layer it lands on.
"""
from __future__ import annotations

import os
import re
from typing import Iterable, List, Sequence, Tuple

import torch
import torch.nn as nn

UNET_REPLACE = ("UNet3DConditionModel",)
TEXT_ENCODER_REPLACE = ("CLIPEncoderLayer",)

# layers that are plain torch.nn classes in diffusers 0.24's UNet3D building blocks (see module docstring)
_PLAIN_024 = re.compile(r"^(conv_in|conv_in2|conv_out|time_embedding\.cond_proj|motion_embedding\.[02]|"
                        r"(.+\.)?temp_convs\.\d+\.conv[1-4]\.\d+|(.+\.)?temp_attentions\.\d+\.proj_(in|out)|"
                        r"transformer_in\.proj_(in|out))$")


def _candidates(model: nn.Module, replace_modules: Iterable[str]) -> List[Tuple[str, nn.Module]]:
    """Every Linear / Conv2d / Conv3d below a module whose class name is in `replace_modules`, in the reference's
    traversal order (`_find_modules_v2`, utils/lora.py:269-313)."""
    names = set(replace_modules)
    out, seen = [], set()
    for anc_name, anc in model.named_modules():
        if anc.__class__.__name__ not in names:
            continue
        for full, m in anc.named_modules():
            if isinstance(m, (nn.Linear, nn.Conv2d, nn.Conv3d)) and id(m) not in seen:
                seen.add(id(m))
                out.append(((anc_name + "." if anc_name else "") + full, m))
    return out


def lora_targets(model: nn.Module, n_pairs: int, replace_modules: Iterable[str] = UNET_REPLACE, mode: str = "auto"):
    """The layers an adapter file with `n_pairs` (up, down) pairs addresses."""
    every = _candidates(model, replace_modules)
    plain = [(n, m) for n, m in every if _PLAIN_024.match(n)]
    if mode == "all" or (mode == "auto" and n_pairs == len(every)):
        return every
    if mode == "plain" or (mode == "auto" and n_pairs == len(plain)):
        return plain
    raise ValueError(f"LoRA file holds {n_pairs} adapter pairs; the model has {len(every)} Linear/Conv layers under "
                     f"{sorted(replace_modules)} ({len(plain)} of them plain torch.nn layers in diffusers 0.24): "
                     "cannot tell which layers the adapters belong to")


@torch.no_grad()
def fold_lora_(model: nn.Module, loras: Sequence[torch.Tensor], replace_modules: Iterable[str] = UNET_REPLACE,
               scale: float = 1.0, mode: str = "auto") -> List[str]:
    """W += scale * up @ down for every addressed layer, in place (fp32 arithmetic, stored in the layer's dtype).
    Returns the names of the folded layers.  In-place updates bump the parameters' version counters, which is what the
    packed-weight caches of the HIP layers key on; captured hipGraphs are dropped through `invalidate_caches`."""
    loras = list(loras)
    if len(loras) % 2:
        raise ValueError("LoRA list must hold (up, down) pairs")
    targets = lora_targets(model, len(loras) // 2, replace_modules, mode)
    # every pair is checked BEFORE the first weight is touched: a bad file leaves the model as it was
    for (name, m), up, down in zip(targets, loras[0::2], loras[1::2]):
        w = m.weight
        r = up.shape[1]
        if up.shape[0] != w.shape[0] or down.shape[0] != r or tuple(down.shape[1:]) != tuple(w.shape[1:]) or \
                any(s != 1 for s in up.shape[2:]):
            raise ValueError(f"LoRA pair {tuple(up.shape)} x {tuple(down.shape)} does not fit layer {name} {tuple(w.shape)}")
    done = []
    try:
        for (name, m), up, down in zip(targets, loras[0::2], loras[1::2]):
            w = m.weight
            delta = up.to(w.device, torch.float32).flatten(1) @ down.to(w.device, torch.float32).flatten(1)
            w.add_((scale * delta).reshape(w.shape).to(w.dtype))
            done.append(name)
    finally:
        if hasattr(model, "invalidate_caches"):
            model.invalidate_caches()
    return done


def inject_inferable_lora(model, lora_path="", unet_replace_modules=("UNet3DConditionModel",),
                          text_encoder_replace_modules=("CLIPEncoderLayer",), is_extended=False, r=16, scale=1.0):
    """Signature of the reference's `inject_inferable_lora` (utils/lora.py:482-526): `model` is the pipeline; every
    `*.pt` in `lora_path` whose name contains "unet" / "text_encoder" is folded into that component.  (`r` is implied by
    the tensors; kept for signature compatibility.)  Returns {component: [layer names]}."""
    out = {}
    if not lora_path or not os.path.exists(lora_path):
        return out
    for f in sorted(os.listdir(lora_path)):
        if not f.endswith(".pt"):
            continue
        path = os.path.join(lora_path, f)
        if "text_encoder" in f and getattr(model, "text_encoder", None) is not None:
            out["text_encoder"] = fold_lora_(model.text_encoder, torch.load(path, map_location="cpu"),
                                             text_encoder_replace_modules, scale, mode="all")
            print("Successfully loaded Text Encoder LoRa.")
        elif "unet" in f and getattr(model, "unet", None) is not None:
            out["unet"] = fold_lora_(model.unet, torch.load(path, map_location="cpu"), unet_replace_modules, scale)
            print("Successfully loaded UNET LoRa.")
        else:
            print("Found a .pt file, but doesn't have the correct name format. (unet.pt, text_encoder.pt)")
    return out
