"""`train.py --eval`-style driver end to end on the GPU: a synthetic diffusers-layout checkpoint is written with
save_pretrained, then animate_anything_amd.eval loads it (from_pretrained), encodes an image, noises, denoises,
decodes and writes the GIF - the reference flow of train.py:731-857."""
import json
import os

import numpy as np
import pytest
import torch
import yaml
from PIL import Image

from animate_anything_amd import eval as aa_eval
from animate_anything_amd.unet3d import UNet3DConditionModel
from animate_anything_amd.vae import AutoencoderKL
from util import SMALL_UNET, SMALL_VAE

pytestmark = pytest.mark.gpu


def test_eval_driver_roundtrip(tmp_path):
    torch.manual_seed(0)
    ckpt = tmp_path / "ckpt"
    unet = UNet3DConditionModel(**SMALL_UNET)
    unet.save_pretrained(str(ckpt / "unet"))
    AutoencoderKL(**SMALL_VAE).save_pretrained(str(ckpt / "vae"))
    os.makedirs(ckpt / "scheduler")
    json.dump({"_class_name": "DDIMScheduler", "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
               "num_train_timesteps": 1000, "steps_offset": 1, "timestep_spacing": "leading"},
              open(ckpt / "scheduler" / "scheduler_config.json", "w"))
    # reloading gives identical weights and the config survives
    again = UNet3DConditionModel.from_pretrained(str(ckpt), subfolder="unet")
    assert again.config.block_out_channels == tuple(SMALL_UNET["block_out_channels"])
    assert all(torch.equal(a, b) for a, b in zip(unet.state_dict().values(), again.state_dict().values()))

    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 255, (120, 160, 3), dtype=np.uint8)).save(tmp_path / "img.jpg")
    m = np.zeros((120, 160), dtype=np.uint8)
    m[30:90, 40:120] = 255
    Image.fromarray(m).save(tmp_path / "img_label.jpg")
    torch.save({"prompt_embeds": torch.randn(1, 77, 128), "negative_prompt_embeds": torch.randn(1, 77, 128)},
               tmp_path / "embeds.pt")
    cfg = {"pretrained_model_path": str(ckpt), "motion_mask": True, "motion_strength": True, "seed": 7,
           "output_dir": str(tmp_path / "out"), "iters": 2,
           "validation_data": {"prompt": "", "prompt_image": str(tmp_path / "img.jpg"), "mask": str(tmp_path / "img_label.jpg"),
                               "prompt_embeds": str(tmp_path / "embeds.pt"), "num_frames": 4, "width": 128, "height": 128,
                               "num_inference_steps": 10, "guidance_scale": 9, "fps": 8}}
    yaml.safe_dump(cfg, open(tmp_path / "config.yaml", "w"))
    results = aa_eval.main(["--config", str(tmp_path / "config.yaml"), "--eval", "validation_data.num_inference_steps=3"])
    assert len(results) == 2
    _, frames, latents = results[0]
    # 160x120 image at a 128x128 budget keeps its aspect ratio, rounded to multiples of 8 (train.py:741-744)
    assert frames[0].shape == (112, 144, 3) and len(frames) == 4
    assert latents.shape == (1, 4, 4, 14, 18) and torch.isfinite(latents).all()
    assert os.path.exists(tmp_path / "out" / "img" / "0.gif") and os.path.exists(tmp_path / "out" / "img" / "1.gif")
    assert Image.open(tmp_path / "out" / "img" / "0.gif").n_frames == 4

    # ---- the same sample through the CPU oracle (reference flow train.py:731-791 restated with oracle modules): same
    # checkpoint, same image / mask preprocessing, the SAME initial noise (first draw of the seeded GPU generator), same
    # prompt embeddings, motion strength index+3, DPM-Solver++ timesteps -> final latents must agree (fp16 bar: MSE < 1e-3)
    import oracle
    from safetensors.torch import load_file
    ounet = oracle.UNet3DConditionModel(**SMALL_UNET).eval()
    ounet.load_state_dict(load_file(str(ckpt / "unet" / "diffusion_pytorch_model.safetensors")))
    ovae = oracle.AutoencoderKL(**SMALL_VAE).eval()
    ovae.load_state_dict(load_file(str(ckpt / "vae" / "diffusion_pytorch_model.safetensors")))
    h, w, frames_n, steps = 112, 144, 4, 3
    img = aa_eval.preprocess_image(Image.open(tmp_path / "img.jpg").convert("RGB"), h, w)
    with torch.no_grad():
        x0 = oracle.tensor_to_vae_latent(img[None], ovae)                                    # [1,4,1,14,18]
    g = torch.Generator(device="cuda").manual_seed(7)
    noise = torch.randn((1, 4, frames_n, h // 8, w // 8), dtype=torch.float16, device="cuda", generator=g).float().cpu()
    osched = oracle.DPMSolverMultistepScheduler()
    osched.set_timesteps(steps)
    init = oracle.ddpm_add_noise(x0.repeat(1, 1, frames_n, 1, 1), noise, int(osched.timesteps[0]))
    np_mask = np.array(Image.open(tmp_path / "img_label.jpg").resize((w, h)))
    np_mask[np_mask != 0] = 255
    mask = aa_eval.mask_to_latent(np_mask, h // 8, w // 8)
    emb = torch.load(tmp_path / "embeds.pt")
    _, want = oracle.LatentToVideoPipeline(ovae, ounet, osched)(
        latents=init, prompt_embeds=emb["prompt_embeds"], negative_prompt_embeds=emb["negative_prompt_embeds"],
        condition_latent=x0, mask=mask, motion=[3], num_inference_steps=steps, guidance_scale=9.0, return_dict=False,
        timesteps=osched.timesteps)
    mse = ((latents.float().cpu() - want) ** 2).mean().item()
    assert mse < 1e-3, mse


def test_stage2_rgba_eval_driver_roundtrip(tmp_path):
    """train_transparent_i2v_stage2.py --eval flow (RGBA image -> premultiplied VAE latent + alpha offset -> UNet3D denoising
    -> VAE decode -> UNet384 alpha decode -> gif / webp) on synthetic small checkpoints."""
    from animate_anything_amd import eval_stage2
    from animate_anything_amd.layerdiffuse import LatentTransparencyOffsetEncoder, UNet384
    torch.manual_seed(0)
    ckpt, alpha_ckpt = tmp_path / "unet_ckpt", tmp_path / "alpha_ckpt"
    UNet3DConditionModel(**SMALL_UNET).save_pretrained(str(ckpt / "unet"))
    AutoencoderKL(**SMALL_VAE).save_pretrained(str(ckpt / "vae"))
    os.makedirs(ckpt / "scheduler")
    json.dump({"_class_name": "DDIMScheduler", "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
               "num_train_timesteps": 1000, "steps_offset": 1, "timestep_spacing": "leading"},
              open(ckpt / "scheduler" / "scheduler_config.json", "w"))
    os.makedirs(alpha_ckpt)
    enc, dec = LatentTransparencyOffsetEncoder(), UNet384()
    with torch.no_grad():
        for p_ in list(enc.parameters()) + list(dec.parameters()):
            if p_.abs().max() == 0:
                p_.normal_(0.0, 0.05)
    torch.save(enc.state_dict(), alpha_ckpt / "vae_alpha_encoder.pth")
    torch.save(dec.state_dict(), alpha_ckpt / "vae_alpha_decoder.pth")
    rng = np.random.default_rng(1)
    rgba = rng.integers(0, 255, (128, 128, 4), dtype=np.uint8)
    rgba[..., 3] = 0
    rgba[32:96, 40:100, 3] = 255
    Image.fromarray(rgba, "RGBA").save(tmp_path / "obj.png")
    torch.save({"prompt_embeds": torch.randn(1, 77, 128), "negative_prompt_embeds": torch.randn(1, 77, 128)}, tmp_path / "embeds.pt")
    cfg = {"transparent_unet_pretrained_model_path": str(ckpt), "transparent_VAE_pretrained_model_path": str(alpha_ckpt),
           "motion_mask": True, "motion_strength": True, "seed": 3, "output_dir": str(tmp_path / "out"), "iters": 1,
           "validation_data": {"prompt": "", "prompt_image": str(tmp_path / "obj.png"), "prompt_embeds": str(tmp_path / "embeds.pt"),
                               "num_frames": 3, "width": 128, "height": 128, "num_inference_steps": 2, "guidance_scale": 9, "fps": 6}}
    yaml.safe_dump(cfg, open(tmp_path / "config2.yaml", "w"))
    results = eval_stage2.main(["--config", str(tmp_path / "config2.yaml"), "--eval"])
    assert len(results) == 1
    _, frames, latents, pngs, alpha = results[0]
    assert len(frames) == 3 and frames[0].shape == (128, 128, 3)
    assert latents.shape == (1, 4, 3, 16, 16) and torch.isfinite(latents).all()
    assert pngs.shape == (3, 128, 128, 4) and alpha.shape == (3, 128, 128) and set(np.unique(alpha).tolist()) <= {0, 255}
    out = tmp_path / "out" / "obj"
    assert (out / "0.gif").exists() and (out / "0_decoded_rgba.webp").exists() and (out / "0_decoded_alpha.webp").exists()
    assert Image.open(out / "0_decoded_rgba.webp").n_frames >= 1
