"""LatentToVideoPipeline on MI355X: drop-in for /root/reference/models/pipeline.py:12-214
(`LatentToVideoPipeline.__call__`, a diffusers `TextToVideoSDPipeline` subclass) and the latent
helpers of /root/reference/utils/common.py:12-48,296-300.

Same keyword arguments, same return convention (`(frames, latents)` when `return_dict=False`,
pipeline.py:211-212).  Per denoising step the device work is: one UNet3D forward on the CFG-doubled
batch (hipGraph replay when enabled) and one fused guidance + DPM-Solver++ update kernel
(`aa_cfg_dpm_step`) -- the reference's cat / chunk / permute / scheduler.step chain
(pipeline.py:165-192) collapses into those two.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .schedulers import DPMSolverMultistepScheduler


class TextToVideoSDPipelineOutput(SimpleNamespace):
    pass


def tensor_to_vae_latent(t, vae):
    """reference utils/common.py:12-20."""
    b, f = t.shape[:2]
    lat = vae.encode(t.reshape((b * f,) + tuple(t.shape[2:]))).latent_dist.mode()
    lat = lat.reshape((b, f) + tuple(lat.shape[1:])).permute(0, 2, 1, 3, 4)
    return lat * 0.18215


def DDPM_forward_timesteps(x0, step, num_frames, scheduler, generator=None):
    """reference utils/common.py:32-48: x0 repeated over frames, noised to timesteps[len-step]."""
    timesteps = scheduler.timesteps[len(scheduler.timesteps) - step:]
    xt = x0.repeat(1, 1, num_frames, 1, 1) if x0.shape[2] == 1 else x0
    noise = torch.randn(xt.shape, dtype=xt.dtype, device=xt.device, generator=generator)
    t = torch.tensor([int(timesteps[0])] * xt.shape[0], device=xt.device)
    return scheduler.add_noise(xt, noise, t), timesteps


def calculate_latent_motion_score(latents):
    """reference utils/common.py:296-300."""
    diff = (latents[:, :, 1:] - latents[:, :, :-1]).abs()
    return diff.mean(dim=[2, 3, 4]).sum(dim=1) * 10


def tensor2vid(video):
    """diffusers 0.24 tensor2vid (mean = std = 0.5): [b,c,f,h,w] -> list of f uint8 [h, b*w, c] frames."""
    video = (video.float() * 0.5 + 0.5).clamp(0, 1)
    b, c, f, h, w = video.shape
    frames = (video.permute(2, 3, 0, 4, 1).reshape(f, h, b * w, c) * 255).to(torch.uint8).cpu().numpy()
    return [fr for fr in frames]


class LatentToVideoPipeline:
    cfg_shared_prefix = True     # compute the text-independent prefix of the UNet once per guidance pair: the same arithmetic per element, but
                                 # other tile shapes / summation orders in the shared part - the two forms differ inside the fp16 noise floor
                                 # (0.013-0.029 of the latents' range over three steps at guidance 9, tests/test_gpu_fullsize.py), not bit for bit
    guidance_group = None        # (role, process group) of distributed.guidance_pair: the two guidance halves on two GPUs

    def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet=None, scheduler=None):
        self.vae, self.text_encoder, self.tokenizer, self.unet = vae, text_encoder, tokenizer, unet
        self.scheduler = scheduler if scheduler is not None else DPMSolverMultistepScheduler()
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8

    @classmethod
    def from_pretrained(cls, path, text_encoder=None, vae=None, unet=None, tokenizer=None, scheduler=None, **_):
        """diffusers directory layout (reference train.py:799-804).  Components passed by the caller
        are used as they are; the scheduler is rebuilt from `scheduler/scheduler_config.json`."""
        from .unet3d import UNet3DConditionModel
        from .vae import AutoencoderKL
        if unet is None:
            unet = UNet3DConditionModel.from_pretrained(path, subfolder="unet")
        if vae is None:
            vae = AutoencoderKL.from_pretrained(path, subfolder="vae")
        if scheduler is None:
            cfg_path = os.path.join(path, "scheduler", "scheduler_config.json")
            cfg = json.load(open(cfg_path)) if os.path.exists(cfg_path) else {}
            scheduler = DPMSolverMultistepScheduler.from_config(cfg)
        if tokenizer is None or text_encoder is None:
            try:                   # the tokenizer is transformers' (host-side string processing), the text tower is clip.py
                from transformers import CLIPTokenizer
                from .clip import CLIPTextModel
                tokenizer = tokenizer or CLIPTokenizer.from_pretrained(path, subfolder="tokenizer")
                text_encoder = text_encoder or CLIPTextModel.from_pretrained(path, subfolder="text_encoder")
            except Exception:      # text encoder is optional: callers may pass prompt_embeds
                pass
        return cls(vae, text_encoder, tokenizer, unet, scheduler)

    def save_pretrained(self, path):
        """diffusers directory layout (what `from_pretrained` reads; reference train.py:298-299): unet/, vae/,
        scheduler/scheduler_config.json, text_encoder/ + tokenizer/ through transformers when present, model_index.json."""
        os.makedirs(path, exist_ok=True)
        index = {"_class_name": "LatentToVideoPipeline"}
        for name in ("unet", "vae", "text_encoder", "tokenizer"):
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
        for m in (self.vae, self.unet, self.text_encoder):
            if m is not None:
                if device is not None:
                    m.to(device)
                if torch_dtype is not None:
                    m.to(torch_dtype)
        return self

    # ------------------------------------------------------------------ helpers (diffusers TextToVideoSDPipeline)
    def check_inputs(self, prompt, height, width, callback_steps, negative_prompt=None, prompt_embeds=None,
                     negative_prompt_embeds=None):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")

    def _encode_prompt(self, prompt, device, do_cfg, negative_prompt=None, prompt_embeds=None,
                       negative_prompt_embeds=None):
        def clip(texts):
            ids = self.tokenizer(texts, padding="max_length", max_length=self.tokenizer.model_max_length,
                                 truncation=True, return_tensors="pt").input_ids.to(device)
            return self.text_encoder(ids)[0]

        if prompt_embeds is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise ValueError("no text encoder loaded: pass `prompt_embeds` / `negative_prompt_embeds`")
            texts = [prompt] if isinstance(prompt, str) else list(prompt)
            prompt_embeds = clip(texts)
        if do_cfg and negative_prompt_embeds is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise ValueError("classifier-free guidance needs `negative_prompt_embeds` when no text encoder is loaded")
            neg = [negative_prompt or ""] * prompt_embeds.shape[0] if not isinstance(negative_prompt, list) else negative_prompt
            negative_prompt_embeds = clip(neg)
        if do_cfg:
            prompt_embeds = torch.cat([negative_prompt_embeds.to(prompt_embeds), prompt_embeds])
        return prompt_embeds

    def decode_latents(self, latents):
        latents = latents / self.vae.config.scaling_factor
        b, c, f, h, w = latents.shape
        img = self.vae.decode(latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)).sample
        return img.reshape((b, f) + tuple(img.shape[1:])).permute(0, 2, 1, 3, 4).float()

    # ------------------------------------------------------------------ denoising
    def denoise(self, latents, prompt_embeds, condition_latent, mask, motion, timesteps, guidance_scale,
                callback=None, callback_steps=1):
        """The hot loop (reference pipeline.py:163-198).  `prompt_embeds` already holds [uncond; text]
        when guidance is on.  Latents are kept in fp32.

        With the DPM-Solver++ scheduler one step is exactly two device submissions: the UNet session (a hipGraph replay
        holding timestep embedding + input packing + the whole forward when graphs are enabled) and the fused
        guidance + solver kernel, which reads the UNet's token-layout output, updates the fp32 latents in place - they ARE
        the session's `sample` input, duplicated for guidance by the packing kernel - and deposits the next timestep."""
        cfg = guidance_scale > 1.0
        sched = self.scheduler
        unet = self.unet
        if not (isinstance(sched, DPMSolverMultistepScheduler) and hasattr(unet, "session")):
            return self._denoise_generic(latents, prompt_embeds, condition_latent, mask, motion, timesteps, guidance_scale,
                                         callback, callback_steps)
        dev = latents.device
        b0, _, frames, h, w = latents.shape
        if cfg and self.guidance_group is not None:
            return self._denoise_guidance_parallel(latents, prompt_embeds, condition_latent, mask, motion, timesteps,
                                                   guidance_scale, callback, callback_steps)
        b = 2 * b0 if cfg else b0
        use_mask = bool(unet.motion_mask and mask is not None)
        has_motion = bool(unet.motion_strength and motion is not None)
        # under guidance both halves of the UNet batch see the same latents / condition frame / mask / timestep / motion and
        # differ only in the text: the UNet computes the text-independent prefix once (UNet3DConditionModel._core cfg_dup)
        shared = cfg and self.cfg_shared_prefix and b0 % condition_latent.shape[0] == 0 and (not use_mask or b0 % mask.shape[0] == 0)
        sess = unet.session(b, frames, h, w, tuple(prompt_embeds.shape[1:]), use_mask, has_motion, False, torch.float32, b0,
                            condition_latent.shape[0], mask.shape[0] if use_mask else 0, dev, cfg_dup=shared)
        mot = None
        if has_motion:
            mot = torch.as_tensor(motion, device=dev).to(torch.float32).reshape(-1).expand(b).contiguous()
        ts = [int(t) for t in timesteps]
        sess.load(sample=latents, cond=condition_latent, mask=mask if use_mask else None, text=prompt_embeds, motion=mot,
                  t=torch.full((b,), float(ts[0]) if ts else 0.0, dtype=torch.float32, device=dev))
        x = sess.inputs["sample"]                               # fp32 [b0,C,T,h,w]: updated in place by the solver kernel
        x0_prev = torch.zeros_like(x)
        for i, t in enumerate(ts):
            eps = sess.run()                                    # tokens [b*(T+1)*h*w, 8], columns [0, out_channels) valid (row pitch = stride(0))
            k = sched.coefficients(sched.index_for_timestep(t), i > 0)
            ops.cfg_dpm_step_tokens(eps, x, x0_prev, None, guidance_scale if cfg else None, k,
                                    next_t=sess.inputs["t"], next_t_value=float(ts[i + 1]) if i + 1 < len(ts) else float(t))
            if callback is not None and i % callback_steps == 0:
                callback(i, t, x)
        return x.clone()

    def _denoise_guidance_parallel(self, latents, prompt_embeds, condition_latent, mask, motion, timesteps, guidance_scale,
                                   callback=None, callback_steps=1):
        """Latency mode: this rank runs ONE half of the guidance batch (role 0: unconditional context = first half of
        `prompt_embeds`, role 1: the text half), the pair exchanges the UNet outputs with one all-gather per step (RCCL over
        xGMI; 0.56 MB per rank at 16x64x64) and both ranks apply the identical guidance + solver update to their copy of the
        latents.  Same arithmetic per element as `denoise` (the two halves of a guidance batch never interact inside the UNet)."""
        from . import distributed as D
        role, group = self.guidance_group
        sched, unet, dev = self.scheduler, self.unet, latents.device
        b0, _, frames, h, w = latents.shape
        use_mask = bool(unet.motion_mask and mask is not None)
        has_motion = bool(unet.motion_strength and motion is not None)
        text = prompt_embeds[role * b0:(role + 1) * b0]
        sess = unet.session(b0, frames, h, w, tuple(text.shape[1:]), use_mask, has_motion, False, torch.float32, b0,
                            condition_latent.shape[0], mask.shape[0] if use_mask else 0, dev, cfg_dup=False)
        mot = None
        if has_motion:
            mot = torch.as_tensor(motion, device=dev).to(torch.float32).reshape(-1).expand(b0).contiguous()
        ts = [int(t) for t in timesteps]
        sess.load(sample=latents, cond=condition_latent, mask=mask if use_mask else None, text=text, motion=mot,
                  t=torch.full((b0,), float(ts[0]) if ts else 0.0, dtype=torch.float32, device=dev))
        x = sess.inputs["sample"]
        x0_prev = torch.zeros_like(x)
        for i, t in enumerate(ts):
            # (the UNet's token matrix is padded to 8 columns: only conv_out's channels travel - 0.28 MB per rank at 16x64x64)
            eps = D.all_gather_cat(sess.run()[:, :unet.conv_out.out_channels], group)          # [uncond clips | text clips], the layout the solver kernel reads
            k = sched.coefficients(sched.index_for_timestep(t), i > 0)
            ops.cfg_dpm_step_tokens(eps, x, x0_prev, None, guidance_scale, k,
                                    next_t=sess.inputs["t"], next_t_value=float(ts[i + 1]) if i + 1 < len(ts) else float(t))
            if callback is not None and i % callback_steps == 0:
                callback(i, t, x)
        return x.clone()

    def _denoise_generic(self, latents, prompt_embeds, condition_latent, mask, motion, timesteps, guidance_scale,
                         callback=None, callback_steps=1):
        """Any scheduler with a diffusers-style `step()`: the UNet through its module `forward`, guidance and the update as
        torch expressions (the reference's own sequence, pipeline.py:165-192)."""
        cfg = guidance_scale > 1.0
        dt = self.unet.dtype
        sched = self.scheduler
        x = latents.float().contiguous()
        cond = torch.cat([condition_latent, condition_latent]) if cfg else condition_latent     # :160-161
        for i, t in enumerate(timesteps):
            x_lp = x.to(dt)
            model_in = torch.cat([x_lp, x_lp]) if cfg else x_lp                                  # :165
            eps = self.unet(model_in, t, encoder_hidden_states=prompt_embeds, condition_latent=cond, mask=mask,
                            motion=None if motion is None else torch.as_tensor(motion, device=x.device)).sample
            e_u, e_t = (eps[: x.shape[0]], eps[x.shape[0]:]) if cfg else (eps, eps)
            e = e_u.float() + guidance_scale * (e_t.float() - e_u.float()) if cfg else eps.float()
            b, c, f, h, w = x.shape
            flat = sched.step(e.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w), t,
                              x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)).prev_sample
            x = flat.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4).contiguous()
            if callback is not None and i % callback_steps == 0:
                callback(i, t, x)
        return x

    @torch.no_grad()
    def __call__(self, prompt=None, height=None, width=None, num_frames=16, num_inference_steps=50,
                 guidance_scale=9.0, negative_prompt=None, eta=0.0, generator=None, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, output_type="np", return_dict=True,
                 callback=None, callback_steps=1, cross_attention_kwargs=None, condition_latent=None,
                 mask=None, timesteps=None, motion=None):
        if latents is None:
            raise ValueError("LatentToVideoPipeline expects caller-prepared `latents` (reference pipeline.py:153)")
        height = height or latents.shape[-2] * self.vae_scale_factor
        width = width or latents.shape[-1] * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps, negative_prompt, prompt_embeds, negative_prompt_embeds)
        device = latents.device
        do_cfg = guidance_scale > 1.0
        prompt_embeds = self._encode_prompt(prompt, device, do_cfg, negative_prompt, prompt_embeds, negative_prompt_embeds)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        if timesteps is None:
            timesteps = self.scheduler.timesteps
        x = self.denoise(latents, prompt_embeds, condition_latent, mask, motion, [int(t) for t in timesteps],
                         guidance_scale, callback, callback_steps)
        latents = x.to(self.unet.dtype)
        if self.vae is None or output_type == "latent":
            video = None
        else:
            video_tensor = self.decode_latents(latents)
            video = video_tensor if output_type == "pt" else tensor2vid(video_tensor)
        if not return_dict:
            return (video, latents)
        return TextToVideoSDPipelineOutput(frames=video)
