"""The reference's Stable-Video-Diffusion pipelines on MI355X: drop-ins for
/root/reference/models/pipeline.py:223-466 (`MaskStableVideoDiffusionPipeline.__call__`) and :468-731
(`TextStableVideoDiffusionPipeline.__call__`), both subclasses of diffusers' `StableVideoDiffusionPipeline` whose helpers
(`_encode_image`, `_encode_vae_image`, `_get_add_time_ids`, `prepare_latents`, `decode_latents`, `check_inputs`,
`image_processor.preprocess`, `tensor2vid`) are restated in `StableVideoDiffusionPipeline` below.

Same keyword arguments and return convention.  Per denoising step the device work is one UNet session run (a hipGraph
replay holding the embeddings, the input assembly `cat([mask, latents / sqrt(sigma^2+1), image_latents], dim=2)` and the
whole forward) and one fused per-frame-guidance + Euler update kernel (`aa_cfg_euler_step_tokens`), which also deposits
the next step's timestep and input scale - the reference's cat / scale_model_input / chunk / scheduler.step chain
(pipeline.py:417-440) collapses into those two.  The CLIP image encoder is `clip.CLIPVisionModelWithProjection` (a `transformers`
module passed by the caller works as well).
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .schedulers import EulerDiscreteScheduler


class StableVideoDiffusionPipelineOutput(SimpleNamespace):
    """diffusers StableVideoDiffusionPipelineOutput: `.frames`."""


def _append_dims(x, target_dims):
    """reference models/pipeline.py:216-221."""
    dims_to_append = target_dims - x.ndim
    if dims_to_append < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * dims_to_append]


def _gaussian_1d(k, s, like):
    xs = torch.arange(k, dtype=torch.float32, device=like.device) - k // 2
    if k % 2 == 0:
        xs = xs + 0.5
    g = torch.exp(-xs.pow(2.0) / (2.0 * s * s))
    return (g / g.sum()).to(like.dtype)


def _resize_with_antialiasing(x, size):
    """diffusers `_resize_with_antialiasing` (pipeline_stable_video_diffusion.py): Gaussian blur sized by the down-scaling
    factors (reflect padding), then bicubic interpolation with align_corners=True.  Host-side preprocessing of ONE image."""
    h, w = x.shape[-2:]
    factors = (h / size[0], w / size[1])
    sigmas = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
    ks = [int(max(2.0 * 2.0 * s, 3)) for s in sigmas]
    ks = [k + 1 if k % 2 == 0 else k for k in ks]
    c = x.shape[1]
    ky, kx = _gaussian_1d(ks[0], sigmas[0], x), _gaussian_1d(ks[1], sigmas[1], x)
    xp = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
    xp = F.conv2d(xp, kx.reshape(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)
    xp = F.conv2d(xp, ky.reshape(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)
    return F.interpolate(xp, size=size, mode="bicubic", align_corners=True)


class VaeImageProcessor:
    """The two diffusers VaeImageProcessor methods the path uses: `preprocess` (PIL / array / tensor -> [B,3,H,W] in [-1,1],
    resized to a multiple of the VAE scale factor) and `postprocess` (denormalise to [0,1], to numpy / PIL)."""

    def __init__(self, vae_scale_factor=8):
        self.vae_scale_factor = vae_scale_factor

    def preprocess(self, image, height=None, width=None):
        if torch.is_tensor(image):
            x = image if image.dim() == 4 else image[None]
            if height is not None and (x.shape[-2] != height or x.shape[-1] != width):
                x = F.interpolate(x.float(), size=(height, width)).to(x.dtype)
            return x if x.min() < 0 else 2.0 * x - 1.0
        imgs = image if isinstance(image, list) else [image]
        out = []
        for im in imgs:
            if hasattr(im, "resize"):                           # PIL
                if height is not None:
                    im = im.resize((width, height), resample=1)     # PIL LANCZOS (VaeImageProcessor resample="lanczos")
                a = np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0
            else:
                a = np.asarray(im, dtype=np.float32)
            out.append(torch.from_numpy(a).permute(2, 0, 1))
        return 2.0 * torch.stack(out) - 1.0

    @staticmethod
    def postprocess(image, output_type="pil"):
        x = (image.float() / 2 + 0.5).clamp(0, 1)
        if output_type == "pt":
            return x
        a = x.cpu().permute(0, 2, 3, 1).numpy()
        if output_type == "np":
            return a
        from PIL import Image
        return [Image.fromarray((f * 255).round().astype("uint8")) for f in a]


def tensor2vid(video, processor, output_type="np"):
    """diffusers pipeline_stable_video_diffusion.tensor2vid: [B,C,F,H,W] -> per clip the F post-processed frames."""
    outputs = []
    for b in range(video.shape[0]):
        outputs.append(processor.postprocess(video[b].permute(1, 0, 2, 3), output_type))
    if output_type == "np":
        return np.stack(outputs)
    if output_type == "pt":
        return torch.stack(outputs)
    return outputs


class StableVideoDiffusionPipeline:
    """The diffusers base class, as far as the reference's two subclasses use it."""

    fused_step = True     # False: the reference's own per-step sequence (module forward, torch guidance, scheduler.step)

    def __init__(self, vae=None, image_encoder=None, unet=None, scheduler=None, feature_extractor=None):
        self.vae, self.image_encoder, self.unet, self.feature_extractor = vae, image_encoder, unet, feature_extractor
        self.scheduler = scheduler if scheduler is not None else EulerDiscreteScheduler()
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8
        self.image_processor = VaeImageProcessor(self.vae_scale_factor)
        self._guidance_scale = None

    @classmethod
    def from_pretrained(cls, path, torch_dtype=None, variant=None, vae=None, unet=None, image_encoder=None, scheduler=None, **_):
        """diffusers directory layout (reference train_svd.py:85-91)."""
        from .svd_unet import UNetSpatioTemporalConditionModel
        from .svd_vae import AutoencoderKLTemporalDecoder
        if unet is None:
            unet = UNetSpatioTemporalConditionModel.from_pretrained(path, subfolder="unet", torch_dtype=torch_dtype, variant=variant)
        if vae is None:
            vae = AutoencoderKLTemporalDecoder.from_pretrained(path, subfolder="vae", torch_dtype=torch_dtype, variant=variant)
        if scheduler is None:
            cfg_path = os.path.join(path, "scheduler", "scheduler_config.json")
            scheduler = EulerDiscreteScheduler.from_config(json.load(open(cfg_path)) if os.path.exists(cfg_path) else {})
        feature_extractor = None   # (`_encode_image` applies the CLIP resize / normalisation itself)
        if image_encoder is None:
            try:
                from .clip import CLIPVisionModelWithProjection
                image_encoder = CLIPVisionModelWithProjection.from_pretrained(path, subfolder="image_encoder", torch_dtype=torch_dtype, variant=variant)
            except Exception:      # optional: callers may pass image_embeddings
                pass
        return cls(vae, image_encoder, unet, scheduler, feature_extractor)

    def save_pretrained(self, path):
        """diffusers directory layout (what `from_pretrained` reads; reference train_svd.py:103,320-321): unet/, vae/,
        scheduler/scheduler_config.json, image_encoder/ + feature_extractor/ through transformers when present."""
        os.makedirs(path, exist_ok=True)
        index = {"_class_name": "StableVideoDiffusionPipeline"}
        for name in ("unet", "vae", "image_encoder", "feature_extractor"):
            m = getattr(self, name)
            if m is not None and hasattr(m, "save_pretrained"):
                m.save_pretrained(os.path.join(path, name))
                index[name] = [type(m).__module__.split(".")[0], type(m).__name__]
        os.makedirs(os.path.join(path, "scheduler"), exist_ok=True)
        with open(os.path.join(path, "scheduler", "scheduler_config.json"), "w") as f:
            json.dump(dict(vars(self.scheduler.config), _class_name=type(self.scheduler).__name__), f, indent=2)
        index["scheduler"] = ["animate_anything_amd", type(self.scheduler).__name__]
        with open(os.path.join(path, "model_index.json"), "w") as f:
            json.dump(index, f, indent=2)

    def to(self, device=None, torch_dtype=None, **_):
        for m in (self.vae, self.unet, self.image_encoder):
            if m is not None:
                if device is not None:
                    m.to(device)
                if torch_dtype is not None:
                    m.to(torch_dtype)
        return self

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def _execution_device(self):
        return self.unet.device

    def check_inputs(self, image, height, width):
        if not torch.is_tensor(image) and not hasattr(image, "resize") and not isinstance(image, list):
            raise ValueError("`image` has to be of type `torch.FloatTensor` or `PIL.Image.Image` or `List[PIL.Image.Image]` but is"
                             f" {type(image)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")

    # CLIP normalisation constants of the SVD feature extractor (preprocessor_config.json)
    _CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
    _CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

    def _encode_image(self, image, device, num_videos_per_prompt, do_classifier_free_guidance, image_embeddings=None):
        """diffusers `_encode_image`: antialiased resize to 224x224, CLIP normalisation, vision tower + projection ->
        [B, 1, D]; the unconditional half is zeros, placed first."""
        if image_embeddings is None:
            if self.image_encoder is None:
                raise ValueError("no image encoder loaded: pass `image_embeddings`")
            dtype = next(self.image_encoder.parameters()).dtype
            x = image
            if not torch.is_tensor(x):                         # (a tensor is taken as CLIP-ready, like diffusers does)
                x = _resize_with_antialiasing(self.image_processor.preprocess(x).float(), (224, 224))
                x = (x + 1.0) / 2.0
                mean = torch.tensor(self._CLIP_MEAN).reshape(1, 3, 1, 1)
                std = torch.tensor(self._CLIP_STD).reshape(1, 3, 1, 1)
                x = (x - mean.to(x)) / std.to(x)
            image_embeddings = self.image_encoder(x.to(device=device, dtype=dtype)).image_embeds.unsqueeze(1)
        e = image_embeddings
        bs, seq, _ = e.shape
        e = e.repeat(1, num_videos_per_prompt, 1).reshape(bs * num_videos_per_prompt, seq, -1)
        if do_classifier_free_guidance:
            e = torch.cat([torch.zeros_like(e), e])
        return e

    def _encode_vae_image(self, image, device, num_videos_per_prompt, do_classifier_free_guidance):
        """diffusers `_encode_vae_image`: latent_dist.mode() (NOT scaled); the unconditional half is zeros, placed first."""
        lat = self.vae.encode(image.to(device)).latent_dist.mode()
        # The reference upcasts a half-precision VAE to fp32 for this call when `config.force_upcast` is set (pipeline.py:373-383):
        # the point is range, not precision - fp16 activations of the encoder can overflow at 576 x 1024.  The HIP kernels keep
        # half-precision STORAGE with fp32 accumulation, so the guard here is: fp16 first; if anything overflowed, once more with
        # bf16 storage (fp32's exponent range).  One host sync per clip.
        if (self.vae.dtype == torch.float16 and getattr(self.vae.config, "force_upcast", False) and not bool(torch.isfinite(lat).all())):
            # the retry runs on a bf16 COPY of the encoder half: the live fp16 parameters are never touched (an in-place
            # fp16 -> bf16 -> fp16 round trip would leave every weight rounded to 8 mantissa bits for the rest of the pipeline's life
            # and re-pack every weight cache twice)
            import copy
            enc = copy.copy(self.vae)                           # shallow: shares config and the (unused here) decoder
            enc._modules = dict(self.vae._modules)
            enc._modules["encoder"] = copy.deepcopy(self.vae.encoder).to(torch.bfloat16)
            enc._modules["quant_conv"] = copy.deepcopy(self.vae.quant_conv).to(torch.bfloat16)
            lat = enc.encode(image.to(device, torch.bfloat16)).latent_dist.mode().to(torch.float16)
        if do_classifier_free_guidance:
            lat = torch.cat([torch.zeros_like(lat), lat])
        return lat.repeat(num_videos_per_prompt, 1, 1, 1)

    def _get_add_time_ids(self, fps, motion_bucket_id, noise_aug_strength, dtype, batch_size, num_videos_per_prompt,
                          do_classifier_free_guidance):
        ids = [fps, motion_bucket_id, noise_aug_strength]
        cfg = self.unet.config
        passed = cfg.addition_time_embed_dim * len(ids)
        expected = self.unet.add_embedding.linear_1.in_features
        if expected != passed:
            raise ValueError(f"Model expects an added time embedding vector of length {expected}, but a vector of {passed} was created. "
                             "The model has an incorrect config. Please check `unet.config.time_embedding_type` and "
                             "`text_encoder_2.config.projection_dim`.")
        t = torch.tensor([ids], dtype=dtype).repeat(batch_size * num_videos_per_prompt, 1)
        return torch.cat([t, t]) if do_classifier_free_guidance else t

    def prepare_latents(self, batch_size, num_frames, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_frames, num_channels_latents // 2, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            latents = torch.randn(shape, generator=generator, device=device, dtype=dtype)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    def decode_latents(self, latents, num_frames, decode_chunk_size=14):
        """[B, F, C, h, w] -> [B, 3, F, H, W] fp32, `decode_chunk_size` frames per VAE call."""
        z = latents.flatten(0, 1) / self.vae.config.scaling_factor
        frames = []
        for i in range(0, z.shape[0], decode_chunk_size):
            chunk = z[i:i + decode_chunk_size]
            frames.append(self.vae.decode(chunk, num_frames=chunk.shape[0]).sample)
        frames = torch.cat(frames)
        return frames.reshape((-1, num_frames) + tuple(frames.shape[1:])).permute(0, 2, 1, 3, 4).float()

    # ------------------------------------------------------------------ denoising
    def denoise(self, latents, image_embeddings, added_time_ids, condition_latent, mask, guidance_scale, timesteps,
                callback=None):
        """The hot loop (reference pipeline.py:413-451, :679-716).  latents [B, F, 4, h, w] (already times
        init_noise_sigma); image_embeddings / added_time_ids / condition_latent hold [uncond; cond] when guidance is on;
        mask [Bm, F, 1, h, w] or None; guidance_scale: fp32 [F] per-frame scale or None.  Latents are kept in fp32."""
        sched, unet = self.scheduler, self.unet
        dev = latents.device
        dt = unet.dtype
        b0, frames, c, h, w = latents.shape
        cfg = guidance_scale is not None
        b = 2 * b0 if cfg else b0
        use_mask = mask is not None and unet.config.in_channels == 9
        if unet.config.in_channels == 9 and mask is None:
            raise ValueError("the UNet has the motion-mask input channel (in_channels == 9): `mask` is required")
        n_in = (1 if use_mask else 0) + c + condition_latent.shape[2]
        if n_in != unet.config.in_channels:
            raise ValueError(f"mask / latents / condition latents add up to {n_in} channels, the UNet expects {unet.config.in_channels}")
        if not (self.fused_step and isinstance(sched, EulerDiscreteScheduler) and hasattr(unet, "session")):
            return self._denoise_generic(latents, image_embeddings, added_time_ids, condition_latent, mask if use_mask else None,
                                         guidance_scale, timesteps, callback)
        srcs, vals = [], []
        if use_mask:
            srcs.append((mask.shape[0], 1, dt)); vals.append(mask.to(dt))
        srcs.append((b0, c, torch.float32)); vals.append(latents.float())
        scaled = len(srcs) - 1
        srcs.append((condition_latent.shape[0], condition_latent.shape[2], dt)); vals.append(condition_latent.to(dt))
        sess = unet.session(b, frames, h, w, tuple(image_embeddings.shape[1:]), tuple(srcs), dev, scaled_src=scaled)
        ts = [float(t) for t in timesteps]
        i0 = sched.index_for_timestep(ts[0]) if ts else 0
        sess.load(t=torch.full((b,), ts[0] if ts else 0.0, dtype=torch.float32, device=dev),
                  ids=added_time_ids.to(torch.float32), text=image_embeddings.to(dt),
                  scale=torch.full((1,), sched.input_scale(i0), dtype=torch.float32, device=dev),
                  **{f"src{k}": v for k, v in enumerate(vals)})
        x = sess.inputs[f"src{scaled}"]                          # fp32 [b0,F,4,h,w]: updated in place by the solver kernel
        g = guidance_scale.to(device=dev, dtype=torch.float32).contiguous() if cfg else None
        for i, t in enumerate(ts):
            v = sess.run()                                       # tokens [b*F*h*w, ld >= out_channels] (consumers take the row pitch)
            idx = sched.index_for_timestep(t)
            k = sched.coefficients(idx)
            nxt = i + 1 < len(ts)
            ops.cfg_euler_step_tokens(v, x, g, k["c_x"], k["c_v"], next_t=sess.inputs["t"],
                                      next_t_value=ts[i + 1] if nxt else t, next_scale=sess.inputs["scale"],
                                      next_scale_value=sched.input_scale(sched.index_for_timestep(ts[i + 1])) if nxt else 1.0)
            if callback is not None:
                new = callback(i, t, x)                          # the reference takes `latents` back from the callback (pipeline.py:445-447)
                if new is not None and new is not x:
                    x.copy_(new.to(x.dtype))
        return x.clone()

    def _denoise_generic(self, latents, image_embeddings, added_time_ids, condition_latent, mask, guidance_scale, timesteps,
                         callback=None):
        """Any scheduler with a diffusers-style `step()`: the reference's own sequence as torch expressions around the
        UNet's module `forward` (pipeline.py:415-440)."""
        sched, unet = self.scheduler, self.unet
        dt = unet.dtype
        cfg = guidance_scale is not None
        x = latents.float()
        g = guidance_scale.reshape(1, -1, 1, 1, 1).to(x) if cfg else None
        for i, t in enumerate(timesteps):
            xin = torch.cat([x] * 2) if cfg else x
            xin = sched.scale_model_input(xin, t).to(dt)
            # (the Mask pipeline hands a mask of batch 2 for the guidance pair, also when guidance is off: row b uses mask b % Bm)
            mk = mask.to(dt)[torch.arange(xin.shape[0], device=mask.device) % mask.shape[0]] if mask is not None else None
            parts = ([mk] if mk is not None else []) + [xin, condition_latent.to(dt)]
            v = unet(torch.cat(parts, dim=2), t, encoder_hidden_states=image_embeddings.to(dt), added_time_ids=added_time_ids,
                     return_dict=False)[0].float()
            if cfg:
                vu, vc = v.chunk(2)
                v = vu + g * (vc - vu)
            x = sched.step(v, t, x).prev_sample
            if callback is not None:
                new = callback(i, t, x)
                if new is not None:
                    x = new.to(x.dtype)
        return x

    # ------------------------------------------------------------------ shared body of the two reference __call__s
    def _run(self, image, image_embeddings, height, width, num_frames, num_inference_steps, min_guidance_scale,
             max_guidance_scale, fps, motion_bucket_id, noise_aug_strength, decode_chunk_size, num_videos_per_prompt, generator,
             latents, output_type, callback_on_step_end, return_dict, mask, condition_latent, cfg_mask_cat):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        num_frames = num_frames if num_frames is not None else self.unet.config.num_frames
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else num_frames
        self.check_inputs(image, height, width)
        if hasattr(image, "resize"):
            batch_size = 1
        elif isinstance(image, list):
            batch_size = len(image)
        else:
            batch_size = image.shape[0]
        device = self._execution_device
        do_cfg = max_guidance_scale > 1.0
        dt = self.unet.dtype
        fps = fps - 1                                                                   # :366, :631
        image = self.image_processor.preprocess(image, height=height, width=width).to(device)
        noise = torch.randn(image.shape, generator=generator, device=image.device, dtype=image.dtype)
        image = image + noise_aug_strength * noise                                      # :369-371
        if condition_latent is None:
            lat = self._encode_vae_image(image, device, num_videos_per_prompt, do_cfg).to(image_embeddings.dtype)
            condition_latent = lat.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)          # :384-386
        elif do_cfg:
            condition_latent = torch.cat([condition_latent] * 2)                        # :656-657
        motion_mask = self.unet.config.in_channels == 9
        if mask is not None and motion_mask:
            if cfg_mask_cat:                                                            # Text...: mask arrives [B,F,1,h,w], cat for CFG (:621-622)
                mask5 = torch.cat([mask] * 2) if do_cfg else mask
            else:                                                                       # Mask...: '1 h w -> 2 f 1 h w' (:387)
                mask5 = mask.reshape(1, 1, 1, *mask.shape[-2:]).expand(2, num_frames, 1, *mask.shape[-2:])
            mask5 = mask5.to(device).contiguous()
        else:
            mask5 = None
        added_time_ids = self._get_add_time_ids(fps, motion_bucket_id, noise_aug_strength, image_embeddings.dtype, batch_size,
                                                num_videos_per_prompt, do_cfg).to(device)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        latents = self.prepare_latents(batch_size * num_videos_per_prompt, num_frames, self.unet.config.in_channels, height, width,
                                       image_embeddings.dtype, device, generator, latents)
        guidance = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames).unsqueeze(0)
        guidance = guidance.to(device, latents.dtype).repeat(batch_size * num_videos_per_prompt, 1)
        self._guidance_scale = _append_dims(guidance, latents.ndim)                     # :405-410
        cb = None
        if callback_on_step_end is not None:
            def cb(i, t, x):
                outputs = callback_on_step_end(self, i, t, {"latents": x})
                return outputs.pop("latents", x) if isinstance(outputs, dict) else None
        x = self.denoise(latents, image_embeddings, added_time_ids, condition_latent, mask5,
                         guidance[0].float() if do_cfg else None, timesteps, cb)
        latents = x.to(dt)
        if output_type != "latent":
            frames = self.decode_latents(latents, num_frames, decode_chunk_size)
            frames = tensor2vid(frames, self.image_processor, output_type=output_type)
        else:
            frames = latents
        if not return_dict:
            return frames
        return StableVideoDiffusionPipelineOutput(frames=frames)


class MaskStableVideoDiffusionPipeline(StableVideoDiffusionPipeline):
    """reference models/pipeline.py:223-466."""

    @torch.no_grad()
    def __call__(self, image, height=576, width=1024, num_frames=None, num_inference_steps=25, min_guidance_scale=1.0,
                 max_guidance_scale=3.0, fps=7, motion_bucket_id=127, noise_aug_strength=0.02, decode_chunk_size=None,
                 num_videos_per_prompt=1, generator=None, latents=None, output_type="pil", callback_on_step_end=None,
                 callback_on_step_end_tensor_inputs=("latents",), return_dict=True, mask=None, image_embeddings=None):
        """`image_embeddings` (extension): a ready CLIP image embedding [B, 1, D] instead of running the vision tower."""
        do_cfg = max_guidance_scale > 1.0
        emb = self._encode_image(image, self._execution_device, num_videos_per_prompt, do_cfg, image_embeddings)   # :360
        return self._run(image, emb.to(self.unet.dtype), height, width, num_frames, num_inference_steps, min_guidance_scale,
                         max_guidance_scale, fps, motion_bucket_id, noise_aug_strength, decode_chunk_size, num_videos_per_prompt,
                         generator, latents, output_type, callback_on_step_end, return_dict, mask, None, cfg_mask_cat=False)


class TextStableVideoDiffusionPipeline(StableVideoDiffusionPipeline):
    """reference models/pipeline.py:468-731."""

    @torch.no_grad()
    def __call__(self, image, prompt_embeds=None, negative_prompt_embeds=None, height=576, width=1024, num_frames=None,
                 num_inference_steps=25, min_guidance_scale=1.0, max_guidance_scale=3.0, fps=7, motion_bucket_id=127,
                 noise_aug_strength=0.02, decode_chunk_size=None, num_videos_per_prompt=1, generator=None, latents=None,
                 output_type="pil", callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",),
                 return_dict=True, mask=None, condition_type="image", condition_latent=None, image_embeddings=None):
        do_cfg = max_guidance_scale > 1.0
        dev = self._execution_device
        if condition_type == "image":                                                   # :606-607
            emb = self._encode_image(image, dev, num_videos_per_prompt, do_cfg, image_embeddings)
        elif condition_type == "text":                                                  # :608-611
            emb = torch.cat([negative_prompt_embeds, prompt_embeds]) if do_cfg else prompt_embeds
        else:                                                                           # :612-616
            emb = self._encode_image(image, dev, num_videos_per_prompt, do_cfg, image_embeddings)
            pe = torch.cat([negative_prompt_embeds, prompt_embeds]) if do_cfg else prompt_embeds
            emb = torch.cat([emb, pe.to(emb)], dim=1)
        return self._run(image, emb.to(self.unet.dtype), height, width, num_frames, num_inference_steps, min_guidance_scale,
                         max_guidance_scale, fps, motion_bucket_id, noise_aug_strength, decode_chunk_size, num_videos_per_prompt,
                         generator, latents, output_type, callback_on_step_end, return_dict, mask, condition_latent,
                         cfg_mask_cat=True)
