"""`train.py --eval` on MI355X: the reference's eval driver (/root/reference/train.py:731-870: `eval`,
`batch_eval`, `main_eval`, CLI) re-stated on top of the HIP-backed modules of this package.

    python -m animate_anything_amd.eval --config <ckpt>/config.yaml --eval validation_data.prompt_image=img.jpg ...

Same YAML keys and dot-list overrides (`pretrained_model_path`, `validation_data.{prompt,prompt_image,mask,strength,
num_frames,width,height,num_inference_steps,guidance_scale,fps}`, `seed`, `motion_mask`, `motion_strength`).
The third-party pieces the reference pulls in and that do not exist in this image are replaced by small
equivalents: OmegaConf -> PyYAML + dot-list merge, `VaeImageProcessor.preprocess` -> PIL lanczos resize + [-1,1]
scaling, torchvision `ToTensor`/`Resize(antialias=False)` -> torch bilinear interpolate, imageio -> PIL GIF writer.
`calculate_motion_precision` (OpenCV metric on the decoded frames, utils/common.py:88-141) is restated with numpy +
scipy.ndimage (host side, outside the hot path).  If the checkpoint has no loadable CLIP text encoder, `validation_data.prompt_embeds`
(a .pt file with `prompt_embeds` / `negative_prompt_embeds`) may stand in for the prompt.
"""
from __future__ import annotations

import argparse
import json
import math
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F
import yaml
from PIL import Image

from .pipeline import (DDPM_forward_timesteps, LatentToVideoPipeline, calculate_latent_motion_score,
                       tensor_to_vae_latent)
from .schedulers import DDPMScheduler, DPMSolverMultistepScheduler
from .unet3d import UNet3DConditionModel
from .vae import AutoencoderKL


# ----------------------------------------------------------------------------- config (OmegaConf subset)
class Config(dict):
    """dict with attribute access and `.get`, enough for the reference's `validation_data.xxx` usage."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Config({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def load_config(path, dotlist=()):
    """OmegaConf.load + OmegaConf.from_dotlist + merge (train.py:859-867)."""
    with open(path) as f:
        cfg = yaml.safe_load(f) or {}
    for item in dotlist:
        key, _, val = item.partition("=")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val) if val != "" else None
    return _wrap(cfg)


# ----------------------------------------------------------------------------- image helpers
def preprocess_image(pimg, height, width):
    """diffusers 0.24 `VaeImageProcessor.preprocess`: lanczos resize, /255, 2x-1 -> [1,3,H,W]."""
    pimg = pimg.resize((width, height), resample=Image.LANCZOS)
    x = torch.from_numpy(np.asarray(pimg, dtype=np.float32) / 255.0).permute(2, 0, 1)[None]
    return 2.0 * x - 1.0


def load_motion_mask(path, width, height):
    """train.py:750-756: the mask image resized to the pixel size with every non-zero value set to 255 (JPEG ringing around the
    region counts as "moving"), or all-255 without a mask.  uint8 [height, width]."""
    if path:
        np_mask = np.array(Image.open(path).resize((width, height)))
        if np_mask.ndim == 3:
            np_mask = np_mask[..., 0]
        np_mask[np_mask != 0] = 255
        return np_mask
    return np.ones([height, width], dtype=np.uint8) * 255


def mask_to_latent(np_mask, h, w):
    """T.ToTensor()(np_mask) then T.Resize([h,w], antialias=False) (train.py:761-764) -> [1,1,1,h,w]."""
    m = torch.from_numpy(np_mask.astype(np.float32) / 255.0)[None, None]
    m = F.interpolate(m, size=(h, w), mode="bilinear", align_corners=False, antialias=False)
    return m[:, :, None]


def get_moved_area_mask(frames, move_th=5, th=-1):
    """reference utils/common.py:88-133 without OpenCV: frames -> gray (cv2's 8-bit BGR2GRAY fixed-point weights on channels
    0/1/2 - the reference hands it RGB frames as they are), |gray_0 - gray_i| > move_th accumulated over the frames, then the
    union of the bounding rectangles of the connected regions (cv2.findContours is 8-connected; the bounding rectangle of an
    inner contour lies inside its outer one) whose area reaches `th` (default 0.5 % of the frame)."""
    from scipy import ndimage

    def gray(f):
        f = np.asarray(f).astype(np.int64)
        return ((f[..., 0] * 1868 + f[..., 1] * 9617 + f[..., 2] * 4899 + 8192) >> 14).astype(np.int16)

    ref = gray(frames[0])
    total = np.zeros(ref.shape, dtype=bool)
    for f in frames[1:]:
        total |= np.abs(ref - gray(f)) > move_th
    labels, _ = ndimage.label(total, structure=np.ones((3, 3), dtype=bool))
    mask = np.zeros(ref.shape, dtype=np.uint8)
    if th < 0:
        th = int(mask.shape[0] * mask.shape[1] * 0.005)
    for sl in ndimage.find_objects(labels):
        h, w = sl[0].stop - sl[0].start, sl[1].stop - sl[1].start
        if w * h < th:
            continue
        mask[sl] = 255
    return mask


def calculate_motion_precision(frames, mask):
    """reference utils/common.py:136-141: share of the moved area that lies inside the motion mask."""
    moved = get_moved_area_mask(frames, move_th=20, th=0) == 255
    gt = np.asarray(mask) == 255
    return float(np.sum(moved & gt) / np.sum(moved)) if moved.any() else float("nan")


def save_gif(path, frames, fps):
    imgs = [Image.fromarray(f) for f in frames]
    imgs[0].save(path, save_all=True, append_images=imgs[1:], duration=int(1000 / fps), loop=0)


# ----------------------------------------------------------------------------- model loading (train.py:85-104)
def load_primary_models(pretrained_model_path, motion_mask=None, motion_strength=None):
    """reference train.py:85-104.  As there, `unet/config.json` of the checkpoint decides whether the mask input conv
    (`conv_in2`) and the motion-strength embedding are active (`from_pretrained(subfolder="unet")` without overrides,
    train.py:89); `motion_mask` / `motion_strength` override the checkpoint only when given (not None)."""
    sched_cfg = os.path.join(pretrained_model_path, "scheduler", "scheduler_config.json")
    scfg = json.load(open(sched_cfg)) if os.path.exists(sched_cfg) else {}
    noise_scheduler = DDPMScheduler(**{k: v for k, v in scfg.items() if not k.startswith("_")})
    tokenizer = text_encoder = None
    try:
        from transformers import CLIPTokenizer
        from .clip import CLIPTextModel
        tokenizer = CLIPTokenizer.from_pretrained(pretrained_model_path, subfolder="tokenizer")
        text_encoder = CLIPTextModel.from_pretrained(pretrained_model_path, subfolder="text_encoder")
    except Exception:
        pass
    vae = AutoencoderKL.from_pretrained(pretrained_model_path, subfolder="vae")
    overrides = {k: v for k, v in (("motion_mask", motion_mask), ("motion_strength", motion_strength)) if v is not None}
    unet = UNet3DConditionModel.from_pretrained(pretrained_model_path, subfolder="unet", **overrides)
    return noise_scheduler, tokenizer, text_encoder, vae, unet, scfg


# ----------------------------------------------------------------------------- eval (train.py:731-791)
def eval(pipeline, validation_data, out_file, index, forward_t=25, preview=True, generator=None):
    vae = pipeline.vae
    device, dtype = vae.device, vae.dtype
    pimg = Image.open(validation_data.prompt_image)
    pimg = pimg.convert("RGB")
    width, height = pimg.size
    scale = math.sqrt(width * height / (validation_data.height * validation_data.width))
    validation_data.height = round(height / scale / 8) * 8
    validation_data.width = round(width / scale / 8) * 8
    input_image = preprocess_image(pimg, validation_data.height, validation_data.width)
    input_image = input_image.unsqueeze(0).to(dtype).to(device)
    input_image_latents = tensor_to_vae_latent(input_image, vae)

    np_mask = load_motion_mask(validation_data.mask if "mask" in validation_data else None, validation_data.width, validation_data.height)
    Image.fromarray(np_mask).save(os.path.splitext(out_file)[0] + "_mask.jpg")

    initial_latents, timesteps = DDPM_forward_timesteps(input_image_latents, forward_t, validation_data.num_frames,
                                                        pipeline.scheduler, generator=generator)
    b, c, f, h, w = initial_latents.shape
    mask = mask_to_latent(np_mask, h, w).to(dtype).to(device)
    motion_strength = validation_data.get("strength", index + 3)
    prompt_kwargs = dict(prompt=validation_data.get("prompt"))
    if validation_data.get("prompt_embeds"):
        emb = torch.load(validation_data.prompt_embeds, map_location=device)
        prompt_kwargs = dict(prompt_embeds=emb["prompt_embeds"].to(dtype),
                             negative_prompt_embeds=emb["negative_prompt_embeds"].to(dtype))
    with torch.no_grad():
        video_frames, video_latents = pipeline(
            latents=initial_latents, width=validation_data.width, height=validation_data.height,
            num_frames=validation_data.num_frames, num_inference_steps=validation_data.num_inference_steps,
            guidance_scale=validation_data.guidance_scale, condition_latent=input_image_latents, mask=mask,
            motion=[motion_strength], return_dict=False, timesteps=timesteps, **prompt_kwargs)
    if preview:
        save_gif(out_file, video_frames, validation_data.get("fps", 8))
    real_motion_strength = calculate_latent_motion_score(video_latents.float()).cpu().numpy()[0]
    precision = calculate_motion_precision(video_frames, np_mask)                       # train.py:786
    print(f"save file {out_file}, motion strength {motion_strength} -> {real_motion_strength}, motion precision {precision}")
    return precision, video_frames, video_latents


def batch_eval(unet, text_encoder, vae, tokenizer, scheduler_config, validation_data, output_dir, preview,
               global_step=0, iters=6, generator=None, indices=None, seed=None, lora_path=None, lora_rank=16,
               unet_lora_modules=("UNet3DConditionModel",), guidance_group=None):
    """train.py:793-823: pipeline with DPM-Solver++ built from the checkpoint's scheduler config, `iters` samples.
    `indices` (clip sharding, main_eval): render only these sample indices, each with its own generator seeded
    `seed + index` (distributed.clip_seed) so that a sample does not depend on which rank renders it."""
    unet.eval()
    scheduler = DPMSolverMultistepScheduler.from_config(scheduler_config)
    pipeline = LatentToVideoPipeline(vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet,
                                     scheduler=scheduler)
    pipeline.guidance_group = guidance_group              # latency mode: the guidance halves of a clip on the two GPUs of a pair
    if lora_path:                                         # train_lora.py:909-917: adapters for inference (folded, not wrapped)
        from .lora import inject_inferable_lora
        inject_inferable_lora(pipeline, lora_path, r=lora_rank, unet_replace_modules=unet_lora_modules)
        print(f"LoRA injected to {list(unet_lora_modules)}, lora path: {lora_path}")
    scheduler.set_timesteps(validation_data.num_inference_steps, device=vae.device)
    results = []
    from .distributed import clip_seed
    for t in (range(iters) if indices is None else indices):
        if indices is not None and seed is not None:
            generator = torch.Generator(device=vae.device).manual_seed(clip_seed(seed, t))
        name = os.path.basename(validation_data.prompt_image)
        out_dir = f"{output_dir}/{name.split('.')[0]}"
        os.makedirs(out_dir, exist_ok=True)
        out_file = f"{out_dir}/{global_step + t}.gif"
        results.append(eval(pipeline, validation_data, out_file, t, forward_t=validation_data.num_inference_steps,
                            preview=preview, generator=generator))
    return results


def main_eval(pretrained_model_path, validation_data, seed=None, motion_mask=None, motion_strength=None,
              output_dir="output/demo", iters=6, dtype="fp16", graph=True, lora_path=None, lora_rank=16,
              unet_lora_modules=("UNet3DConditionModel",), guidance_parallel=False, **kwargs):
    """train.py:825-857.  Weights are cast to half precision on the GPU ("cuda" is the HIP device on ROCm).
    The reference accepts `motion_mask` / `motion_strength` here and never forwards them to the UNet constructor
    (train.py:838: the checkpoint's config.json governs); so do we - they are passed on only when the YAML sets them.

    Under `torchrun` (WORLD_SIZE > 1: BASELINE.json configs[2], "bs=8, 1 clip/GPU") the `iters` samples of the clip batch
    are sharded round-robin over the ranks (rank r renders samples r, r+W, ...; per-sample seed = seed + index so the
    outputs do not depend on W) and the final latents are all-gathered over RCCL (distributed.gather_clips)."""
    from . import distributed as D
    rank, world, _dev = D.init()
    generator = None
    if seed is not None:
        torch.manual_seed(seed)
        generator = torch.Generator(device="cuda").manual_seed(seed)
    _, tokenizer, text_encoder, vae, unet, scfg = load_primary_models(pretrained_model_path, motion_mask, motion_strength)
    vae.enable_slicing()
    weight_dtype = torch.half if dtype == "fp16" else torch.bfloat16
    for m in (text_encoder, unet, vae):
        if m is not None:
            m.requires_grad_(False)
            m.to(torch.device("cuda"), dtype=weight_dtype)
    if graph:
        unet.enable_graph()
    if world == 1:
        return batch_eval(unet, text_encoder, vae, tokenizer, scfg, validation_data, output_dir, True, iters=iters,
                          generator=generator, lora_path=lora_path, lora_rank=lora_rank, unet_lora_modules=unet_lora_modules)
    if guidance_parallel:
        # latency mode (`guidance_parallel=true`): ranks (2p, 2p+1) share one clip - unconditional / text half of its guidance
        # batch - and exchange the UNet outputs every step; the clips are sharded over the pairs; rank 2p writes the files
        pair, role, group = D.guidance_pair(rank, world)
        return batch_eval(unet, text_encoder, vae, tokenizer, scfg, validation_data, output_dir, role == 0, iters=iters,
                          indices=D.clip_indices(iters, pair, world // 2), seed=seed if seed is not None else 0,
                          lora_path=lora_path, lora_rank=lora_rank, unet_lora_modules=unet_lora_modules,
                          guidance_group=(role, group))
    mine = D.clip_indices(iters, rank, world)
    results = batch_eval(unet, text_encoder, vae, tokenizer, scfg, validation_data, output_dir, True, iters=iters,
                         generator=generator, indices=mine, seed=seed, lora_path=lora_path, lora_rank=lora_rank,
                         unet_lora_modules=unet_lora_modules)
    local = torch.stack([r[2][0] for r in results]) if results else None
    shape = [0] * 5
    if local is not None:
        shape = list(local.shape)
    import torch.distributed as dist
    t = torch.tensor(shape, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # ranks without a clip learn the per-clip shape
    if local is None:
        local = torch.zeros([0] + t.tolist()[1:], dtype=weight_dtype, device="cuda")
    gathered = D.gather_clips(local.contiguous(), iters, rank, world)        # [iters, 4, frames, h, w] on every rank
    return results, gathered


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default="./configs/my_config.yaml")
    parser.add_argument("--eval", action="store_true")
    parser.add_argument("rest", nargs=argparse.REMAINDER)
    args = parser.parse_args(argv)
    cfg = load_config(args.config, args.rest)
    if not args.eval:
        raise SystemExit("animate_anything_amd implements the --eval (inference) path only; training is out of scope")
    return main_eval(**cfg)


if __name__ == "__main__":
    main()
