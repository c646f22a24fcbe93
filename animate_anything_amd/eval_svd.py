"""`train_svd.py --eval` on MI355X: the reference's Stable-Video-Diffusion eval driver
(/root/reference/train_svd.py:726-826: `eval`, `main_eval`, CLI; model loading :85-91) on top of the HIP-backed modules.

    python -m animate_anything_amd.eval_svd --config example/train_svd_mask.yaml --eval \\
        pretrained_model_path=<ckpt> validation_data.prompt_image=img.jpg

Same YAML keys and dot-list overrides as the reference (`pretrained_model_path`, `validation_data.{prompt_image,prompt,
width,height,num_frames,num_inference_steps,decode_chunk_size,fps,motion_bucket_id}`, `seed`, `eval_file`).  The image is
resized to the requested area at its own aspect ratio in multiples of 64 (:741-745); `<image>_label.jpg`, when present, is the
motion mask (:747-756); a 9-input-channel UNet selects the mask pipeline (:759-778).  OmegaConf / torchvision / imageio are
replaced as in eval.py.  Under `torchrun` the `iters` samples are sharded round-robin over the ranks (distributed.py).
"""
from __future__ import annotations

import argparse
import json
import math
import os

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from .eval import load_config, save_gif
from .svd_pipeline import MaskStableVideoDiffusionPipeline, StableVideoDiffusionPipeline


def load_primary_models(pretrained_model_path, eval=False):
    """reference train_svd.py:85-91."""
    pipeline = MaskStableVideoDiffusionPipeline.from_pretrained(pretrained_model_path,
                                                                torch_dtype=torch.float16 if eval else None)
    return (pipeline, None, pipeline.feature_extractor, pipeline.scheduler, pipeline.image_processor, pipeline.image_encoder,
            pipeline.vae, pipeline.unet)


def convert_svd(pretrained_model_path, out_path):
    """reference train_svd.py:93-103: turn a plain Stable-Video-Diffusion checkpoint (8 input channels) into the motion-mask
    form - a 9-input-channel UNet whose extra (first) input channel has zero conv_in weights, everything else copied."""
    from .svd_unet import UNetSpatioTemporalConditionModel
    pipeline = StableVideoDiffusionPipeline.from_pretrained(pretrained_model_path)
    cfg = dict(vars(pipeline.unet.config), in_channels=9)
    unet = UNetSpatioTemporalConditionModel.from_config(cfg)
    state = {k: v for k, v in pipeline.unet.state_dict().items() if k != "conv_in.weight"}
    missing, unexpected = unet.load_state_dict(state, strict=False)
    assert list(missing) == ["conv_in.weight"] and not unexpected
    with torch.no_grad():
        unet.conv_in.weight.zero_()
        unet.conv_in.weight[:, 1:] = pipeline.unet.conv_in.weight
    unet.invalidate_caches()
    new_pipeline = StableVideoDiffusionPipeline(pipeline.vae, pipeline.image_encoder, unet, pipeline.scheduler,
                                                pipeline.feature_extractor)
    new_pipeline.save_pretrained(out_path)
    return new_pipeline


def eval(pipeline, vae_processor, validation_data, out_file, index, forward_t=25, preview=True, generator=None):
    """reference train_svd.py:726-790."""
    vae = pipeline.vae
    device, dtype = vae.device, vae.dtype
    pipeline.scheduler.set_timesteps(validation_data.num_inference_steps, device=device)
    pimg = Image.open(validation_data.prompt_image)
    if pimg.mode == "RGBA":
        pimg = pimg.convert("RGB")
    width, height = pimg.size
    scale = math.sqrt(width * height / (validation_data.height * validation_data.width))
    block_size = 64
    validation_data.height = round(height / scale / block_size) * block_size
    validation_data.width = round(width / scale / block_size) * block_size

    mask_path = validation_data.prompt_image.split(".")[0] + "_label.jpg"
    if os.path.exists(mask_path):
        mask = Image.open(mask_path).resize((validation_data.width, validation_data.height))
        np_mask = np.array(mask)
        if np_mask.ndim == 3:
            np_mask = np_mask[:, :, 0]
        np_mask[np_mask != 0] = 255
    else:
        np_mask = np.ones([validation_data.height, validation_data.width], dtype=np.uint8) * 255
    Image.fromarray(np_mask).save(os.path.splitext(out_file)[0] + "_mask.jpg")
    motion_mask = pipeline.unet.config.in_channels == 9

    common = dict(image=pimg, width=validation_data.width, height=validation_data.height, num_frames=validation_data.num_frames,
                  num_inference_steps=validation_data.num_inference_steps,
                  decode_chunk_size=validation_data.get("decode_chunk_size"), fps=validation_data.fps,
                  motion_bucket_id=validation_data.motion_bucket_id, generator=generator, output_type="np")
    if validation_data.get("image_embeddings"):            # stand-in for a missing CLIP vision tower (a .pt file [1,1,D])
        common["image_embeddings"] = torch.load(validation_data.image_embeddings, map_location=device).to(dtype)
    with torch.no_grad():
        if motion_mask:
            h, w = validation_data.height // pipeline.vae_scale_factor, validation_data.width // pipeline.vae_scale_factor
            m = torch.from_numpy(np_mask.astype(np.float32) / 255.0)[None, None]                 # T.ToTensor()
            m = F.interpolate(m, size=(h, w), mode="bilinear", align_corners=False, antialias=False)[0]   # T.Resize(antialias=False)
            out = MaskStableVideoDiffusionPipeline.__call__(pipeline, mask=m.to(dtype).to(device), **common)
        else:
            out = MaskStableVideoDiffusionPipeline.__call__(pipeline, **common)
    video_frames = (out.frames[0] * 255).round().astype(np.uint8)          # [F, H, W, 3]
    if preview:
        save_gif(out_file, list(video_frames), validation_data.get("fps", 8))
    return 0, video_frames


def main_eval(pretrained_model_path, validation_data, seed=None, eval_file=None, output_dir="output/svd_out", iters=5,
              graph=True, **kwargs):
    """reference train_svd.py:792-825."""
    from . import distributed as D
    rank, world, _dev = D.init()
    pipeline, _, _, _, vae_processor, _, vae, unet = load_primary_models(pretrained_model_path, eval=True)
    pipeline.to(torch.device("cuda"))
    if graph:
        unet.enable_graph()
    if eval_file is not None:
        eval_list = json.load(open(eval_file))
    else:
        eval_list = [[validation_data.prompt_image, validation_data.get("prompt")]]
    results = []
    for example in eval_list:
        name, prompt = example
        for t in D.clip_indices(iters, rank, world):
            generator = None
            if seed is not None:
                generator = torch.Generator(device="cuda").manual_seed(D.clip_seed(seed, t))
            out_file_dir = f"{output_dir}/{os.path.basename(name).split('.')[0]}"
            os.makedirs(out_file_dir, exist_ok=True)
            out_file = f"{out_file_dir}/{t}.gif"
            validation_data.prompt_image = name
            validation_data.prompt = prompt
            results.append(eval(pipeline, vae_processor, validation_data, out_file, t, generator=generator))
            print("save file", out_file)
    return results


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
