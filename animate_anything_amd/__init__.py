"""animate_anything_amd: MI355X (gfx950) implementation of animate-anything's denoising hot path.

Public surface mirrors the reference's (`UNet3DConditionModel`, `AutoencoderKL`,
`LatentToVideoPipeline`); the arithmetic runs in hand-written HIP kernels
reached through the C ABI of `include/aa_mi355.h` (`libaa_mi355.so`, loaded with ctypes).
"""
__version__ = "0.1.0"
