"""RGBA (layerdiffuse) eval driver on MI355X: the `in_channels == 5` flow of
/root/reference/train_transparent_i2v_stage2.py (`eval` :356-552, `batch_eval` :555-617, `main_eval` :619-660, model loading
:108-130) re-stated on top of the HIP-backed modules - BASELINE.json configs[4].

    python -m animate_anything_amd.eval_stage2 --config cfg.yaml --eval validation_data.prompt_image=apple.png ...

An RGBA image is split into colour and alpha; the VAE encodes the PREMULTIPLIED colour, `LatentTransparencyOffsetEncoder` adds its
offset (:400-426); the UNet3D denoises exactly as in stage 1; the VAE decodes and `UNet384` recovers (foreground, alpha) per frame
(models/pipeline_stage2.py:290-318).  Outputs: `<n>.gif` (premultiplied video), `<n>_decoded_rgba.webp`, `<n>_decoded_alpha.webp`,
`<n>_mask.jpg`.  Host-side stand-ins for absent third-party pieces are those of animate_anything_amd.eval (PyYAML, PIL).
The `in_channels == 9` (ConcatLatentToVideoPipeline) branch of the reference raises TypeError as shipped and
is not reproduced; video inputs (`.mp4` / `.gif` prompt_image, "default False" in the reference) are not read.
"""
from __future__ import annotations

import argparse
import math
import os

import numpy as np
import torch
from PIL import Image

from . import eval as stage1
from .layerdiffuse import LatentTransparencyOffsetEncoder, MaskedLatentToVideoPipeline, UNet384, encode_rgba
from .pipeline import DDPM_forward_timesteps, calculate_latent_motion_score
from .schedulers import DPMSolverMultistepScheduler


def load_alpha_models(alpha_checkpoint=None):
    """train_transparent_i2v_stage2.py:115-128: `vae_alpha_encoder.pth` / `vae_alpha_decoder.pth` state dicts."""
    enc, dec = LatentTransparencyOffsetEncoder(), UNet384()
    if alpha_checkpoint:
        enc.load_state_dict(torch.load(os.path.join(alpha_checkpoint, "vae_alpha_encoder.pth"), map_location="cpu"))
        dec.load_state_dict(torch.load(os.path.join(alpha_checkpoint, "vae_alpha_decoder.pth"), map_location="cpu"))
        print(f"vae_alpha from ckpt {alpha_checkpoint} loaded..")
    return enc, dec


def split_rgba(path):
    """:378-385: an RGBA image -> (RGB image, alpha as an 8-bit grey image)."""
    pimg = Image.open(path)
    if pimg.mode != "RGBA":
        raise ValueError(f"{path}: the transparent pipeline expects an RGBA image (mode {pimg.mode})")
    r, g, b, a = pimg.split()
    return Image.merge("RGB", (r, g, b)), Image.fromarray(np.array(a.convert("L"), dtype=np.uint8))


def preprocess_alpha(alpha_img, height, width):
    """diffusers `VaeImageProcessor.preprocess` of a single-channel image -> [1,1,H,W] in [-1,1]."""
    a = alpha_img.resize((width, height), resample=Image.LANCZOS)
    x = torch.from_numpy(np.asarray(a, dtype=np.float32) / 255.0)[None, None]
    return 2.0 * x - 1.0


def save_anim(path, frames, fps, fmt):
    imgs = [Image.fromarray(f) for f in frames]
    imgs[0].save(path, format=fmt, save_all=True, append_images=imgs[1:], duration=int(1000 / fps), loop=0)


def eval(pipeline, vae_alpha_encoder, vae_alpha_decoder, validation_data, out_file, index, forward_t=25, preview=True,
         generator=None):
    """train_transparent_i2v_stage2.py:356-552, in_channels == 5."""
    vae = pipeline.vae
    device, dtype = vae.device, vae.dtype
    pipeline.scheduler.set_timesteps(validation_data.num_inference_steps, device=device)
    num_frames = validation_data.num_frames
    pimg, pimg_alpha = split_rgba(validation_data.prompt_image)
    width, height = pimg.size
    scale = math.sqrt(width * height / (validation_data.height * validation_data.width))
    block = 64                                                                           # :389
    validation_data.height = round(height / scale / block) * block
    validation_data.width = round(width / scale / block) * block
    image = stage1.preprocess_image(pimg, validation_data.height, validation_data.width).to(dtype).to(device)
    alpha = preprocess_alpha(pimg_alpha, validation_data.height, validation_data.width).to(dtype).to(device)
    latents = encode_rgba(vae, vae_alpha_encoder, image, alpha, num_frames)                # :400-424
    clean_latents = latents.detach().clone()

    stem = validation_data.prompt_image.split(".")[0]
    mask_path = next((p for p in (stem + "_label.jpg", stem + "_label.png") if os.path.exists(p)), None)
    if mask_path is not None:
        np_mask = np.array(Image.open(mask_path).resize((validation_data.width, validation_data.height)))
        if np_mask.ndim == 3:
            np_mask = np_mask[:, :, 0]
        np_mask[np_mask != 0] = 255
    else:
        np_mask = np.ones([validation_data.height, validation_data.width], dtype=np.uint8) * 255
    Image.fromarray(np_mask).save(os.path.splitext(out_file)[0] + "_mask.jpg")
    b, c, f, h, w = latents.shape
    mask_1_frame = stage1.mask_to_latent(np_mask, h, w).to(dtype).to(device)              # [1,1,1,h,w]  (:445-447)

    initial_latents, timesteps = DDPM_forward_timesteps(latents, forward_t, num_frames, pipeline.scheduler, generator=generator)
    motion_strength = index * 2 + 3                                                       # :460
    prompt_kwargs = dict(prompt=validation_data.get("prompt"))
    if validation_data.get("prompt_embeds"):
        emb = torch.load(validation_data.prompt_embeds, map_location=device)
        prompt_kwargs = dict(prompt_embeds=emb["prompt_embeds"].to(dtype), negative_prompt_embeds=emb["negative_prompt_embeds"].to(dtype))
    with torch.no_grad():
        video_frames, video_latents, pngs, alpha_png, pngs_rgb = pipeline(
            clean_latents=clean_latents, vae_alpha_decoder=vae_alpha_decoder, latents=initial_latents,
            width=validation_data.width, height=validation_data.height, num_frames=num_frames,
            num_inference_steps=validation_data.num_inference_steps, guidance_scale=validation_data.guidance_scale,
            motion=[motion_strength], return_dict=False, condition_latent=latents[:, :, :1].detach().clone(), mask=mask_1_frame,
            timesteps=timesteps, **prompt_kwargs)
    if preview:
        fps = validation_data.get("fps", 6)
        save_anim(out_file, video_frames, fps, "GIF")
        save_anim(out_file.replace(".gif", "_decoded_rgba.webp"), list(pngs), fps, "WEBP")
        save_anim(out_file.replace(".gif", "_decoded_alpha.webp"), list(alpha_png), fps, "WEBP")
    real_motion_strength = calculate_latent_motion_score(video_latents.float()).cpu().numpy()[0]
    print(f"save file {out_file}, motion strength {motion_strength} -> {real_motion_strength}")
    err = float(real_motion_strength - motion_strength)
    return err * err, video_frames, video_latents, pngs, alpha_png


def batch_eval(unet, text_encoder, vae, vae_alpha_encoder, vae_alpha_decoder, tokenizer, scheduler_config, validation_data,
               output_dir, preview, examples, global_step=0, iters=3, generator=None):
    """:555-617: `iters` samples (motion strength 3, 5, 7, ...) for every (rgba image, prompt) example."""
    unet.eval()
    pipeline = MaskedLatentToVideoPipeline(vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet,
                                           scheduler=DPMSolverMultistepScheduler.from_config(scheduler_config))
    os.makedirs(output_dir, exist_ok=True)
    results = []
    for name, prompt in examples:
        for t in range(iters):
            out_dir = f"{output_dir}/{os.path.basename(name).split('.')[0]}"
            os.makedirs(out_dir, exist_ok=True)
            validation_data.prompt_image, validation_data.prompt = name, prompt
            results.append(eval(pipeline, vae_alpha_encoder, vae_alpha_decoder, validation_data, f"{out_dir}/{global_step + t}.gif", t,
                                forward_t=validation_data.num_inference_steps, preview=preview, generator=generator))
    return results


def main_eval(validation_data, seed=None, motion_mask=None, motion_strength=None, iters=3, output_dir="output/stage_2_eval",
              transparent_unet_pretrained_model_path="./output/latent/transparent_unet",
              transparent_VAE_pretrained_model_path="./output/latent/transparent_VAE", examples=None, graph=True, **kwargs):
    """:619-660.  `examples`: [[rgba png, prompt], ...]; default = the image / prompt of `validation_data` (the reference
    hard-codes example/example_padded_rgba_pngs/{apple,ziyan0}.png, :575-580)."""
    generator = None
    if seed is not None:
        torch.manual_seed(seed)
        generator = torch.Generator(device="cuda").manual_seed(seed)
    _, tokenizer, text_encoder, vae, unet, scfg = stage1.load_primary_models(transparent_unet_pretrained_model_path, motion_mask,
                                                                             motion_strength)
    enc, dec = load_alpha_models(transparent_VAE_pretrained_model_path)
    vae.enable_slicing()
    for m in (text_encoder, unet, vae, enc, dec):
        if m is not None:
            m.requires_grad_(False)
            m.to(torch.device("cuda"), dtype=torch.half)
    if graph:
        unet.enable_graph()
    examples = examples or [[validation_data.prompt_image, validation_data.get("prompt", "")]]
    return batch_eval(unet, text_encoder, vae, enc, dec, tokenizer, scfg, validation_data, output_dir, True, examples, iters=iters,
                      generator=generator)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default="./configs/my_config.yaml")
    parser.add_argument("--eval", action="store_true")
    parser.add_argument("rest", nargs=argparse.REMAINDER)
    args = parser.parse_args(argv)
    cfg = stage1.load_config(args.config, args.rest)
    if not args.eval:
        raise SystemExit("animate_anything_amd implements the --eval (inference) path only; training is out of scope")
    return main_eval(**cfg)


if __name__ == "__main__":
    main()
