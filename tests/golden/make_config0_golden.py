#!/usr/bin/env python
"""Multi-step golden at BASELINE.json configs[0]: the reference's own example image
(/root/reference/example/barbie2.jpg, 600 x 450) through `train.py --eval` semantics (train.py:731-791) at a 256 x 256
target -> 224 x 296 pixels (train.py:741-744), 8 frames, 10 DPM-Solver++ steps, guidance 9, motion strength 3, no mask
file (all-ones mask), the FULL v1.02 architecture with seeded weights (tests/util.py FULL_UNET / seeded_state) and the SD
AutoencoderKL, all on the fp32 CPU oracle.  Stores the preprocessed image, the oracle's condition latent, the noise, the
prompt embeddings (CLIP bypassed: seeded random `prompt_embeds`) and THE LATENTS AFTER EVERY STEP, so the GPU test reports the
drift step by step.  Needs /root/reference for the image only; ~3-4 minutes on 8 cores.
    python tests/golden/make_config0_golden.py   ->  tests/golden/config0_barbie2_8f_256.pt"""
import math
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from util import FULL_UNET, seeded_state  # noqa: E402

FRAMES, STEPS, TARGET, GUIDANCE, STRENGTH = 8, 10, 256, 9.0, 3.0
torch.set_num_threads(os.cpu_count() or 1)

pimg = Image.open("/root/reference/example/barbie2.jpg").convert("RGB")
w0, h0 = pimg.size
scale = math.sqrt(w0 * h0 / (TARGET * TARGET))                       # train.py:741-744
H, W = round(h0 / scale / 8) * 8, round(w0 / scale / 8) * 8
img = np.asarray(pimg.resize((W, H), resample=Image.LANCZOS)).astype(np.float32) / 255.0     # VaeImageProcessor.preprocess [D-0.24]
image = (torch.from_numpy(img).permute(2, 0, 1)[None] * 2.0 - 1.0).half().float()          # (the GPU path sees fp16 pixels)

torch.manual_seed(0)
vae = oracle.AutoencoderKL().eval()
vae.load_state_dict(seeded_state(vae, seed=7))
torch.manual_seed(0)
unet = oracle.UNet3DConditionModel(**FULL_UNET).eval()
unet.load_state_dict(seeded_state(unet))

g = torch.Generator().manual_seed(2024)
with torch.no_grad():
    cond = oracle.tensor_to_vae_latent(image[None], vae) if hasattr(oracle, "tensor_to_vae_latent") else None
if cond is None:
    from oracle.pipeline import tensor_to_vae_latent
    with torch.no_grad():
        cond = tensor_to_vae_latent(image[None], vae)
h, w = cond.shape[-2:]
noise = torch.randn(1, 4, FRAMES, h, w, generator=g)
pos = torch.randn(1, 77, 1024, generator=g).half().float()
neg = torch.randn(1, 77, 1024, generator=g).half().float()
sched = oracle.DPMSolverMultistepScheduler()
sched.set_timesteps(STEPS)
init = oracle.ddpm_add_noise(cond.repeat(1, 1, FRAMES, 1, 1), noise, int(sched.timesteps[0]))   # utils/common.py:32-48 (forward_t = all steps)
mask = torch.ones(1, 1, 1, h, w)
per_step = []
t0 = time.time()


def grab(i, t, lat):
    per_step.append(lat.clone().half())
    print(f"step {i} t={int(t)}  {time.time() - t0:.0f}s  |x|max {lat.abs().max():.3f}", flush=True)


from oracle.pipeline import LatentToVideoPipeline  # noqa: E402
_, final = LatentToVideoPipeline(None, unet, sched)(
    latents=init, prompt_embeds=pos, negative_prompt_embeds=neg, condition_latent=cond, mask=mask, motion=[STRENGTH],
    num_inference_steps=STEPS, guidance_scale=GUIDANCE, return_dict=False, callback=grab, timesteps=sched.timesteps)
out = os.path.join(HERE, "config0_barbie2_8f_256.pt")
torch.save({"image": image.half(), "cond": cond, "noise": noise.half(), "init": init, "pos": pos.half(), "neg": neg.half(),
            "per_step": torch.stack(per_step), "final": final, "height": H, "width": W, "frames": FRAMES, "steps": STEPS,
            "guidance": GUIDANCE, "strength": STRENGTH, "vae_seed": 7, "seconds": time.time() - t0,
            "generator": "tests/golden/make_config0_golden.py (fp32 CPU oracle; image: reference example/barbie2.jpg)"}, out)
print("wrote", out, os.path.getsize(out) // 1024, "KiB")
