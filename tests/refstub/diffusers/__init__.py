"""Stand-in `diffusers` for tests/test_reference_pin.py (see tests/refstub/README.md).  TEST INFRASTRUCTURE ONLY."""
__version__ = "0.24.0+refstub"
from .pipelines.text_to_video_synthesis.pipeline_text_to_video_synth import TextToVideoSDPipeline  # noqa: E402,F401
from .pipelines.stable_video_diffusion.pipeline_stable_video_diffusion import StableVideoDiffusionPipeline  # noqa: E402,F401
