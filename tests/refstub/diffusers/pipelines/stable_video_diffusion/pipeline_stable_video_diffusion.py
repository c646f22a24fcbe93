"""The parts of diffusers-0.24 `StableVideoDiffusionPipeline` that the reference's `MaskStableVideoDiffusionPipeline.__call__` /
`TextStableVideoDiffusionPipeline.__call__` (models/pipeline.py:225-466, 470-731) reach through `self`, restated on top of oracle
pieces so that the reference's OWN call bodies (guidance batching, mask repeat / concat order, per-frame guidance scale, the
upcast bracket around the VAE, callback protocol) run on CPU.  TEST INFRASTRUCTURE ONLY (tests/refstub/README.md).
What is restated here from the published 0.24.0 source (and therefore NOT pinned by running it): `_encode_vae_image`,
`_get_add_time_ids`, `prepare_latents`, `decode_latents`, `check_inputs`, `VaeImageProcessor.preprocess` for tensors."""
import contextlib
from dataclasses import dataclass

import PIL.Image  # noqa: F401  (the real module imports it; the reference's `import PIL` alone does not load the submodule)
import torch

from oracle import svd as OS


@dataclass
class StableVideoDiffusionPipelineOutput:
    frames: object


def tensor2vid(video, processor, output_type="np"):
    raise NotImplementedError("the stub stops at latents: call with output_type='latent'")


class _Bar:
    def update(self, n=1):
        pass


class _ImageProcessor:
    """VaeImageProcessor.preprocess for a [B,3,H,W] tensor that is already in [-1, 1]: returned as it is (diffusers skips the
    normalisation when the tensor has negative values)."""

    def preprocess(self, image, height=None, width=None):
        if not torch.is_tensor(image):
            raise TypeError("the stub pre-processes tensors only")
        return image


class StableVideoDiffusionPipeline:
    def __init__(self, vae, image_encoder, unet, scheduler, feature_extractor=None):
        self.vae, self.image_encoder, self.unet, self.scheduler, self.feature_extractor = vae, image_encoder, unet, scheduler, feature_extractor
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
        self.image_processor = _ImageProcessor()
        self.image_embeddings = None                 # [B, 1, D]: what the CLIP vision tower would give (not under test)
        self.vae_dtype_trace = []                    # dtypes the VAE was moved to (the force_upcast bracket)

    @property
    def _execution_device(self):
        return torch.device("cpu")

    @property
    def guidance_scale(self):
        return self._guidance_scale

    def check_inputs(self, image, height, width):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")

    def _encode_image(self, image, device, num_videos_per_prompt, do_classifier_free_guidance):
        e = self.image_embeddings
        bs, seq, _ = e.shape
        e = e.repeat(1, num_videos_per_prompt, 1).view(bs * num_videos_per_prompt, seq, -1)
        if do_classifier_free_guidance:
            e = torch.cat([torch.zeros_like(e), e])
        return e

    def _encode_vae_image(self, image, device, num_videos_per_prompt, do_classifier_free_guidance):
        lat = self.vae.encode(image.to(device)).latent_dist.mode()
        if do_classifier_free_guidance:
            lat = torch.cat([torch.zeros_like(lat), lat])
        return lat.repeat(num_videos_per_prompt, 1, 1, 1)

    def _get_add_time_ids(self, fps, motion_bucket_id, noise_aug_strength, dtype, batch_size, num_videos_per_prompt, do_classifier_free_guidance):
        ids = torch.tensor([[fps, motion_bucket_id, noise_aug_strength]], dtype=dtype).repeat(batch_size * num_videos_per_prompt, 1)
        return torch.cat([ids, ids]) if do_classifier_free_guidance else ids

    def prepare_latents(self, batch_size, num_frames, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_frames, num_channels_latents // 2, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            latents = torch.randn(shape, generator=generator, dtype=dtype)
        return latents.to(device) * self.scheduler.init_noise_sigma

    def decode_latents(self, latents, num_frames, decode_chunk_size=14):
        return OS.decode_latents(self.vae, latents, num_frames, decode_chunk_size)

    @contextlib.contextmanager
    def progress_bar(self, total=None):
        yield _Bar()

    def maybe_free_model_hooks(self):
        pass
