#!/usr/bin/env python
"""AutoencoderKLTemporalDecoder at BASELINE configs[3] (576x1024): encode of the conditioning image, decode of the 14 frames
(one chunk and in chunks of 7 like example/train_svd_mask.yaml), and a whole MaskStableVideoDiffusionPipeline.__call__
(25 Euler steps + decode) with seeded random weights of the stable-video-diffusion-img2vid architectures."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd.schedulers import EulerDiscreteScheduler  # noqa: E402
from animate_anything_amd.svd_pipeline import MaskStableVideoDiffusionPipeline  # noqa: E402
from animate_anything_amd.svd_unet import UNetSpatioTemporalConditionModel  # noqa: E402
from animate_anything_amd.svd_vae import AutoencoderKLTemporalDecoder  # noqa: E402

dev, dt = "cuda", torch.float16
torch.manual_seed(0)
with torch.device(dev):
    vae = AutoencoderKLTemporalDecoder()
vae = vae.to(dt).eval()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out = {}
with torch.no_grad():
    img = torch.rand(1, 3, 576, 1024, device=dev, dtype=dt) * 2 - 1
    z = torch.randn(14, 4, 72, 128, device=dev, dtype=dt)
    out["vae_encode_1x576x1024_ms"] = round(timed(lambda: vae.encode(img).latent_dist.mode()), 2)
    out["vae_decode_14x576x1024_one_chunk_ms"] = round(timed(lambda: vae.decode(z, num_frames=14).sample), 2)
    out["vae_decode_14x576x1024_chunks_of_7_ms"] = round(timed(lambda: [vae.decode(z[i:i + 7], num_frames=7).sample for i in (0, 7)]), 2)
    if "--pipeline" in sys.argv:
        with torch.device(dev):
            unet = UNetSpatioTemporalConditionModel(in_channels=9, num_frames=14).to(dt).eval()
        unet.enable_graph()
        pipe = MaskStableVideoDiffusionPipeline(vae, None, unet, EulerDiscreteScheduler())
        mask = torch.zeros(1, 72, 128, device=dev, dtype=dt)
        mask[:, 18:54, 32:96] = 1
        emb = torch.randn(1, 1, 1024, device=dev, dtype=dt)

        def clip():
            return pipe(img.float(), height=576, width=1024, num_frames=14, num_inference_steps=25, decode_chunk_size=7, mask=mask,
                        image_embeddings=emb, output_type="pt").frames

        out["mask_svd_pipeline_call_25_steps_14x576x1024_ms"] = round(timed(clip, reps=2), 1)
print(json.dumps(out))
