"""Generate the committed golden vectors: outputs of the CPU oracle on seeded weights and inputs.

The reference itself cannot run in the build container (diffusers==0.24.0 is not installed and the
pretrained checkpoint is absent), so these vectors pin the ORACLE (regression) and give the GPU tests a
container-independent target; they are not outputs of the reference ("parity unpinned").
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from util import TINY_UNET, TINY_VAE, seeded_state, unet_inputs  # noqa: E402


def main():
    torch.manual_seed(0)
    unet = oracle.UNet3DConditionModel(**TINY_UNET).eval()
    unet.load_state_dict(seeded_state(unet))
    out = {}
    for name, (h, w) in {"unet_6x6": (6, 6), "unet_5x7": (5, 7)}.items():
        i = unet_inputs(h=h, w=w, text_len=9)
        with torch.no_grad():
            out[name] = unet(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample.half()
    torch.manual_seed(0)
    vae = oracle.AutoencoderKL(**TINY_VAE).eval()
    vae.load_state_dict(seeded_state(vae))
    x = torch.rand(2, 3, 12, 10, generator=torch.Generator().manual_seed(7)) * 2 - 1
    with torch.no_grad():
        z = vae.encode(x).latent_dist.mode()
        out["vae_latent"] = z.half()
        out["vae_image"] = vae.decode(z).sample.half()
    s = oracle.DPMSolverMultistepScheduler()
    s.set_timesteps(25)
    out["dpm_timesteps_25"] = s.timesteps.clone()
    out["dpm_sigmas_25"] = torch.tensor(s.sigmas, dtype=torch.float32)
    torch.save(out, os.path.join(HERE, "oracle_golden.pt"))
    print({k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
