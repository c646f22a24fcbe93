"""CPU oracle for the reference's LoRA inference path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates /root/reference/utils/lora.py: `LoraInjectedLinear` :33-74, `LoraInjectedConv2d` :77-159, `LoraInjectedConv3d`
:162-237 (forward = base(x) + scale * lora_up(lora_down(x)); lora_down carries the layer's kernel / stride / padding,
lora_up is 1x1) and the injection traversal `_find_modules_v2` :269-313 / `monkeypatch_or_replace_lora_extended` :861-977
(the wrapper REPLACES the layer in its parent, adapters are taken from a flat [up, down, up, down, ...] list in traversal
order).  The product never wraps layers: it folds the adapters into the weights (animate_anything_amd/lora.py); the tests
check fold == injected forward.
"""
import torch
import torch.nn as nn


class LoraInjected(nn.Module):
    def __init__(self, base: nn.Module, up: torch.Tensor, down: torch.Tensor, scale: float = 1.0):
        super().__init__()
        self.base, self.scale = base, scale
        r = up.shape[1]
        if isinstance(base, nn.Linear):
            self.lora_down, self.lora_up = nn.Linear(base.in_features, r, bias=False), nn.Linear(r, base.out_features, bias=False)
        elif isinstance(base, nn.Conv2d):
            self.lora_down = nn.Conv2d(base.in_channels, r, base.kernel_size, base.stride, base.padding, bias=False)
            self.lora_up = nn.Conv2d(r, base.out_channels, 1, bias=False)
        else:
            self.lora_down = nn.Conv3d(base.in_channels, r, base.kernel_size, padding=base.padding, bias=False)
            self.lora_up = nn.Conv3d(r, base.out_channels, 1, bias=False)
        for a in ("in_channels", "out_channels", "in_features", "out_features"):     # (callers of the oracle read these off the layer)
            if hasattr(base, a):
                setattr(self, a, getattr(base, a))
        self.lora_up.weight.data = up.clone().reshape(self.lora_up.weight.shape)
        self.lora_down.weight.data = down.clone().reshape(self.lora_down.weight.shape)

    def forward(self, x):
        return self.base(x) + self.lora_up(self.lora_down(x)) * self.scale


def inject(model: nn.Module, names, loras, scale=1.0):
    """Wrap the layers `names` (dotted paths, traversal order) with the adapters of the flat list `loras`."""
    for name, up, down in zip(names, loras[0::2], loras[1::2]):
        *path, leaf = name.split(".")
        parent = model
        for p in path:
            parent = parent.get_submodule(p)
        parent._modules[leaf] = LoraInjected(parent._modules[leaf], up, down, scale)
    return model
