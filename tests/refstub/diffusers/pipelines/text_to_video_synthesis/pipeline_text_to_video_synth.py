"""The parts of diffusers-0.24 `TextToVideoSDPipeline` that the reference's `LatentToVideoPipeline.__call__`
(models/pipeline.py:12-214) calls on `self`, on top of oracle pieces: enough to run the reference's own denoising loop on CPU."""
import contextlib
from dataclasses import dataclass

import torch

from oracle import pipeline as OP


@dataclass
class TextToVideoSDPipelineOutput:
    frames: object


def tensor2vid(video, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    return OP.tensor2vid(video)


class _Bar:
    def update(self, n=1):
        pass


class TextToVideoSDPipeline:
    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler):
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8

    def check_inputs(self, prompt, height, width, callback_steps, negative_prompt=None, prompt_embeds=None, negative_prompt_embeds=None):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")

    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt=None,
                       prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None):
        if prompt_embeds is None:
            raise ValueError("the stub encodes no text: pass `prompt_embeds`")
        if do_classifier_free_guidance:
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])      # [uncond; text], diffusers order
        return prompt_embeds

    def prepare_extra_step_kwargs(self, generator, eta):
        return {}

    @contextlib.contextmanager
    def progress_bar(self, total=None):
        yield _Bar()

    def decode_latents(self, latents):
        return OP.LatentToVideoPipeline(self.vae, self.unet, self.scheduler).decode_latents(latents)
