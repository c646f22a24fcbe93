"""GPU parity of AutoencoderKL and of the whole LatentToVideoPipeline.__call__ (UNet3D denoising loop +
fused CFG/DPM-Solver++ kernel + VAE decode) against the CPU oracle on identical seeds/inputs.
Metric of record: latent MSE < 1e-3 (north star, fp16)."""
import pytest
import torch

import oracle
from animate_anything_amd.pipeline import LatentToVideoPipeline, tensor_to_vae_latent
from animate_anything_amd.schedulers import DPMSolverMultistepScheduler
from animate_anything_amd.unet3d import UNet3DConditionModel
from animate_anything_amd.vae import AutoencoderKL
from util import SMALL_UNET, SMALL_VAE, rel_err, seeded_state

pytestmark = pytest.mark.gpu


def _vae_pair():
    torch.manual_seed(0)
    ref = oracle.AutoencoderKL(**SMALL_VAE).eval()
    state = seeded_state(ref)
    ref.load_state_dict(state)
    vae = AutoencoderKL(**SMALL_VAE).eval()
    vae.load_state_dict(state)
    return ref, vae.half().cuda()


@pytest.mark.parametrize("h,w", [(64, 64), (88, 72)])
def test_vae_encode_decode(h, w):
    ref, vae = _vae_pair()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, h, w, generator=g) * 2 - 1
    with torch.no_grad():
        want_z = ref.encode(x).latent_dist.mode()
        got_z = vae.encode(x.half().cuda()).latent_dist.mode()
        assert rel_err(got_z, want_z) < 2e-2
        want_img = ref.decode(want_z).sample
        got_img = vae.decode(want_z.half().cuda()).sample
        assert got_img.shape == want_img.shape
        assert rel_err(got_img, want_img) < 2e-2
        # reference helper path: utils/common.py:12-20
        lat = tensor_to_vae_latent(x[None].half().cuda(), vae)
        assert lat.shape == (1, 4, 2, h // 8, w // 8)


@pytest.mark.parametrize("steps", [4])
def test_pipeline_call_latent_mse(steps):
    torch.manual_seed(0)
    ref_unet = oracle.UNet3DConditionModel(**SMALL_UNET).eval()
    state = seeded_state(ref_unet)
    ref_unet.load_state_dict(state)
    unet = UNet3DConditionModel(**SMALL_UNET).eval()
    unet.load_state_dict(state)
    unet = unet.half().cuda()
    ref_vae, vae = _vae_pair()
    g = torch.Generator().manual_seed(4)
    r = lambda *s: torch.randn(*s, generator=g)
    frames, h, w = 3, 12, 12
    x0 = r(1, 4, 1, h, w) * 0.5
    mask = torch.zeros(1, 1, 1, h, w)
    mask[..., 3:9, 3:9] = 1
    pos, neg = r(1, 77, 128), r(1, 77, 128)
    noise = r(1, 4, frames, h, w)
    osched = oracle.DPMSolverMultistepScheduler()
    osched.set_timesteps(steps)
    init = oracle.ddpm_add_noise(x0.repeat(1, 1, frames, 1, 1), noise, int(osched.timesteps[0]))
    want_frames, want_lat = oracle.LatentToVideoPipeline(ref_vae, ref_unet, osched)(
        latents=init, prompt_embeds=pos, negative_prompt_embeds=neg, condition_latent=x0, mask=mask, motion=[4.0],
        num_inference_steps=steps, guidance_scale=9.0, return_dict=False)
    pipe = LatentToVideoPipeline(vae=vae, unet=unet, scheduler=DPMSolverMultistepScheduler())
    dev = lambda t: t.half().cuda()
    got_frames, got_lat = pipe(latents=init.cuda(), prompt_embeds=dev(pos), negative_prompt_embeds=dev(neg),
                               condition_latent=dev(x0), mask=dev(mask), motion=[4.0], num_inference_steps=steps,
                               guidance_scale=9.0, return_dict=False)
    mse = ((got_lat.float().cpu() - want_lat) ** 2).mean().item()
    assert mse < 1e-3, mse
    assert len(got_frames) == frames and got_frames[0].shape == want_frames[0].shape == (h * 8, w * 8, 3)
    diff = sum(abs(a.astype(int) - b.astype(int)).mean() for a, b in zip(got_frames, want_frames)) / frames
    assert diff < 2.0, diff      # mean abs difference in uint8 levels
