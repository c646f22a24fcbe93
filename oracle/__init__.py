"""CPU oracle for the animate-anything denoising hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-torch fp32 *restatement* of the reference algorithm
(`/root/reference/models/unet_3d_condition_mask.py`, `models/unet_3d_blocks.py`,
`models/pipeline.py:12-214`, `utils/common.py:12-48,296-300`) and of the
diffusers==0.24.0 building blocks those files compose (requirements.txt:4; the
dependency is NOT vendored in the reference and is NOT installed here, so its
layers are restated from their published definitions).

PARITY UNPINNED: the reference ships no tests, golden vectors or known-answer
fixtures for this path and neither diffusers nor the
pretrained checkpoint can be loaded in the build container, so this oracle
could not be checked against outputs of the reference itself.  What pins it
instead is listed in DESIGN.md ("Oracle").

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package.  The product package
(`animate_anything_amd`) never does: it fails loudly when the HIP library is
missing rather than fall back to this code.
"""

from .layers import (  # noqa: F401
    sinusoid_embedding,
    TimestepEmbedding,
    ResnetBlock2D,
    TemporalConvLayer,
    Attention,
    FeedForward,
    BasicTransformerBlock,
    Transformer2DModel,
    TransformerTemporalModel,
    Downsample2D,
    Upsample2D,
)
from .unet3d import UNet3DConditionModel, UNet3DConditionOutput  # noqa: F401
from .vae import AutoencoderKL  # noqa: F401
from .scheduler import DPMSolverMultistepScheduler, ddpm_add_noise  # noqa: F401
from .pipeline import LatentToVideoPipeline, tensor2vid, tensor_to_vae_latent  # noqa: F401
