"""Oracle pipeline: CPU restatement of `LatentToVideoPipeline.__call__`
(/root/reference/models/pipeline.py:12-214) and of the latent helpers
/root/reference/utils/common.py:12-20,32-48,296-300 plus the diffusers 0.24
`TextToVideoSDPipeline.decode_latents` / `tensor2vid` they rely on.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

import numpy as np
import torch


def tensor_to_vae_latent(t, vae):
    """utils/common.py:12-20: [b,f,c,h,w] -> vae.encode(...).latent_dist.mode()*0.18215 -> [b,c,f,h,w]."""
    b, f = t.shape[:2]
    lat = vae.encode(t.reshape((b * f,) + t.shape[2:])).latent_dist.mode()
    lat = lat.reshape((b, f) + lat.shape[1:]).permute(0, 2, 1, 3, 4)
    return lat * 0.18215


def ddpm_forward_timesteps(x0, step, num_frames, scheduler, generator=None, noise=None):
    """utils/common.py:32-48: repeat x0 over frames and noise it to timesteps[len-step]."""
    timesteps = scheduler.timesteps[len(scheduler.timesteps) - step:]
    xt = x0.repeat(1, 1, num_frames, 1, 1) if x0.shape[2] == 1 else x0
    if noise is None:
        noise = torch.randn(xt.shape, dtype=xt.dtype, generator=generator)
    t = torch.tensor([int(timesteps[0])] * xt.shape[0])
    return scheduler.add_noise(xt, noise, t), timesteps


def calculate_latent_motion_score(latents):
    """utils/common.py:296-300."""
    diff = (latents[:, :, 1:] - latents[:, :, :-1]).abs()
    return diff.mean(dim=[2, 3, 4]).sum(dim=1) * 10


def tensor2vid(video):
    """diffusers 0.24 `tensor2vid` with the default mean/std 0.5: [b,c,f,h,w] in [-1,1] ->
    list of f uint8 frames of shape [h, b*w, c]."""
    video = (video * 0.5 + 0.5).clamp(0, 1)
    b, c, f, h, w = video.shape
    frames = video.permute(2, 3, 0, 4, 1).reshape(f, h, b * w, c)
    return [(fr.cpu().numpy() * 255).astype(np.uint8) for fr in frames.unbind(0)]


class LatentToVideoPipeline:
    """Drives oracle modules exactly as models/pipeline.py:113-212 drives the reference ones.
    Text encoding is bypassed through `prompt_embeds` / `negative_prompt_embeds`
    (pipeline.py:26-27,136-145); CLIP is outside the hot path."""

    def __init__(self, vae, unet, scheduler, text_encoder=None, tokenizer=None):
        self.vae, self.unet, self.scheduler = vae, unet, scheduler
        self.text_encoder, self.tokenizer = text_encoder, tokenizer
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8

    def decode_latents(self, latents):
        latents = latents / self.vae.config.scaling_factor
        b, c, f, h, w = latents.shape
        z = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        img = self.vae.decode(z).sample
        return img.reshape((b, f) + img.shape[1:]).permute(0, 2, 1, 3, 4).float()

    @torch.no_grad()
    def __call__(self, prompt=None, height=None, width=None, num_frames=16, num_inference_steps=50,
                 guidance_scale=9.0, negative_prompt=None, eta=0.0, generator=None, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, output_type="np", return_dict=True,
                 callback=None, callback_steps=1, cross_attention_kwargs=None, condition_latent=None,
                 mask=None, timesteps=None, motion=None):
        if prompt_embeds is None:
            raise ValueError("the oracle pipeline takes pre-computed `prompt_embeds`")
        height = height or latents.shape[-2] * self.vae_scale_factor
        width = width or latents.shape[-1] * self.vae_scale_factor
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        cfg = guidance_scale > 1.0                                          # pipeline.py:130
        if cfg:
            if negative_prompt_embeds is None:
                raise ValueError("classifier-free guidance needs `negative_prompt_embeds`")
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])   # diffusers _encode_prompt
        self.scheduler.set_timesteps(num_inference_steps)
        if timesteps is None:
            timesteps = self.scheduler.timesteps
        if cfg:
            condition_latent = torch.cat([condition_latent, condition_latent])   # pipeline.py:160-161
        for i, t in enumerate(timesteps):
            x = torch.cat([latents] * 2) if cfg else latents                # pipeline.py:165
            m = torch.tensor(motion) if motion is not None else None
            eps = self.unet(x, t, encoder_hidden_states=prompt_embeds, condition_latent=condition_latent,
                            mask=mask, motion=m).sample
            if cfg:
                e_u, e_t = eps.chunk(2)
                eps = e_u + guidance_scale * (e_t - e_u)                    # pipeline.py:179-181
            b, c, f, h, w = latents.shape
            flat = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
            eflat = eps.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
            flat = self.scheduler.step(eflat, t, flat)                      # pipeline.py:187
            latents = flat.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        if self.vae is None:
            return (None, latents)
        video = self.decode_latents(latents)                                # pipeline.py:200
        frames = video if output_type == "pt" else tensor2vid(video)
        return (frames, latents) if not return_dict else type("Out", (), {"frames": frames})()
