"""End-to-end, multi-step parity at BASELINE.json configs[0] with the FULL architecture:
the reference's example image (example/barbie2.jpg -> 224 x 296 at a 256 x 256 target, train.py:741-744), 8 frames, 10
DPM-Solver++ steps, guidance 9, motion strength 3 - product pipeline on the GPU (fp16 storage: AutoencoderKL encode, UNet3D
denoising loop with the fused guidance + solver kernel, hipGraph on) against the fp32 CPU oracle pipeline
(tests/golden/config0_barbie2_8f_256.pt, generator tests/golden/make_config0_golden.py), latents compared AFTER EVERY STEP.

Tolerance: the north star asks latent MSE < 1e-3 (fp16).  With seeded random weights and guidance 9 the latents grow to
|x| ~ 40 over the ten steps (a trained checkpoint keeps them O(1)), so the bound is applied to the MSE normalised by the
mean square of the oracle latents at that step (equal to the plain MSE for unit-scale latents); the plain MSE is reported.
"""
import json
import os

import pytest
import torch

import oracle
from animate_anything_amd.pipeline import LatentToVideoPipeline, tensor_to_vae_latent
from animate_anything_amd.schedulers import DPMSolverMultistepScheduler
from animate_anything_amd.unet3d import UNet3DConditionModel
from animate_anything_amd.vae import AutoencoderKL
from util import FULL_UNET, rel_err, seeded_state

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_ten_step_pipeline_at_config0_matches_the_oracle_step_by_step():
    gold = torch.load(os.path.join(HERE, "golden", "config0_barbie2_8f_256.pt"))
    torch.manual_seed(0)
    ovae = oracle.AutoencoderKL().eval()
    vae = AutoencoderKL().eval()
    vae.load_state_dict(seeded_state(ovae, seed=gold["vae_seed"]))
    del ovae
    vae = vae.half().cuda()
    torch.manual_seed(0)
    ounet = oracle.UNet3DConditionModel(**FULL_UNET).eval()
    state = seeded_state(ounet)
    del ounet
    unet = UNet3DConditionModel(**FULL_UNET).eval()
    unet.load_state_dict(state)
    del state
    unet = unet.half().cuda()
    unet.enable_graph()

    dev = lambda t: t.half().cuda()
    with torch.no_grad():
        cond = tensor_to_vae_latent(dev(gold["image"])[None], vae)             # utils/common.py:12-20 on the product VAE
    assert cond.shape == gold["cond"].shape
    vae_err = rel_err(cond, gold["cond"])
    assert vae_err < 2e-2, vae_err

    seen = []
    pipe = LatentToVideoPipeline(vae=None, unet=unet, scheduler=DPMSolverMultistepScheduler())
    pipe.scheduler.set_timesteps(gold["steps"])
    with torch.no_grad():
        _, final = pipe(latents=gold["init"].cuda(), prompt_embeds=dev(gold["pos"]), negative_prompt_embeds=dev(gold["neg"]),
                        condition_latent=cond, mask=torch.ones(1, 1, 1, *cond.shape[-2:], device="cuda", dtype=torch.float16),
                        motion=[gold["strength"]], num_inference_steps=gold["steps"], guidance_scale=gold["guidance"],
                        return_dict=False, output_type="latent", timesteps=pipe.scheduler.timesteps,
                        callback=lambda i, t, lat: seen.append(lat.detach().float().cpu().clone()))
    assert len(seen) == gold["steps"]
    rows = []
    for k, (got, want) in enumerate(zip(seen, gold["per_step"].float())):
        mse = ((got - want) ** 2).mean().item()
        scale = (want ** 2).mean().item()
        rows.append({"step": k, "mse": mse, "oracle_mean_square": scale, "normalised_mse": mse / scale,
                     "max_err_over_max": rel_err(got, want)})
    report = {"vae_cond_rel_err": vae_err, "steps": rows}
    print(json.dumps(report, indent=1))
    try:
        os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
        with open(os.path.join(os.path.dirname(HERE), "gpurun_out", "config0_drift.json"), "w") as f:
            json.dump(report, f, indent=1)
    except OSError:
        pass
    for r in rows:
        assert r["normalised_mse"] < 1e-3, r
    assert rel_err(final, gold["final"]) < 5e-2
