#!/usr/bin/env python
"""Golden for the `-m gpu` LoRA test, generated WITH THE REFERENCE'S OWN CODE (needs /root/reference; run in the build
container): /root/reference/utils/lora.py injects trainable adapters into the oracle SMALL_UNET
(`inject_trainable_lora_extended`, :433-479), the zero-initialised up-projections are randomised, `save_lora_weight`
(:569-581) writes the adapter list, and the reference-injected model's forward on the seeded inputs is the expected output.
Adapters are stored in fp16 (the expectation is computed from the rounded values) to keep the fixture small.
    python tests/golden/make_lora_golden.py        ->  tests/golden/lora_small_unet.pt"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
import refload  # noqa: E402
from util import SMALL_UNET, seeded_state, unet_inputs  # noqa: E402

R = refload.load("utils/lora.py")
torch.manual_seed(0)
model = oracle.UNet3DConditionModel(**SMALL_UNET).eval()
model.load_state_dict(seeded_state(model))
R.inject_trainable_lora_extended(model, target_replace_module={"UNet3DConditionModel"}, r=2)
g = torch.Generator().manual_seed(21)
for m in model.modules():
    if isinstance(m, (R.LoraInjectedLinear, R.LoraInjectedConv2d, R.LoraInjectedConv3d)):
        # magnitudes of a trained adapter: the update is a fraction of the base weight (the reference initialises down ~ N(0, 1/r^2)
        # and up = 0; a 0.3 x N(0, 1/r^2) product on every one of 578 layers overflows fp16 activations on the GPU side)
        fan_in = m.lora_down.weight[0].numel()
        m.lora_up.weight.data = (torch.randn(m.lora_up.weight.shape, generator=g) * 0.2).half().float()
        m.lora_down.weight.data = (torch.randn(m.lora_down.weight.shape, generator=g) * 0.2 / fan_in ** 0.5).half().float()
        m.dropout = torch.nn.Identity()
tmp = os.path.join(HERE, "_lora_tmp.pt")
R.save_lora_weight(model, tmp, target_replace_module={"UNet3DConditionModel"})
loras = [t.half() for t in torch.load(tmp)]
os.remove(tmp)
i = unet_inputs(b=2, frames=3, h=8, w=8, text_dim=128)
with torch.no_grad():
    out = model.eval()(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
torch.save({"loras": loras, "expected": out, "inputs": dict(b=2, frames=3, h=8, w=8, text_dim=128), "r": 2,
            "generator": "reference utils/lora.py inject_trainable_lora_extended + save_lora_weight on oracle SMALL_UNET"},
           os.path.join(HERE, "lora_small_unet.pt"))
print(len(loras) // 2, "adapter pairs;", os.path.getsize(os.path.join(HERE, "lora_small_unet.pt")) // 1024, "KiB")
