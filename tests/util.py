"""Shared helpers for the parity tests (seeded weights / inputs of SURVEY.md section 8d)."""
import torch

TINY_UNET = dict(block_out_channels=(64, 128), down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
                 up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"), layers_per_block=1, cross_attention_dim=64,
                 motion_mask=True, motion_strength=True)
SMALL_UNET = dict(block_out_channels=(64, 128, 256, 256), cross_attention_dim=128, motion_mask=True,
                  motion_strength=True)
TINY_VAE = dict(block_out_channels=(32, 64), layers_per_block=1)
SMALL_VAE = dict(block_out_channels=(32, 64, 128, 128))


def seeded_state(module, seed=0, rezero_std=0.02):
    """PyTorch default inits under `seed`; parameters the architecture zero-initialises
    (TemporalConvLayer.conv4, motion_embedding[-1]) are re-drawn N(0, 0.02^2) so no path is vacuous
    (SURVEY.md section 0 item 4)."""
    g = torch.Generator().manual_seed(seed + 1)
    state = {}
    for k, v in module.state_dict().items():
        v = v.clone().float()
        if v.abs().max() == 0:
            v = torch.randn(v.shape, generator=g) * rezero_std
        state[k] = v
    return state


def unet_inputs(b=2, frames=2, h=6, w=6, text_len=77, text_dim=64, seed=1234):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    mask = torch.zeros(1, 1, 1, h, w)
    mask[..., h // 4: h - h // 4, w // 4: w - w // 4] = 1
    return dict(sample=r(b, 4, frames, h, w), cond=r(b, 4, 1, h, w), mask=mask, text=r(b, text_len, text_dim),
                motion=torch.tensor([3.0]), t=501)


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-6)).item()


# ---------------------------------------------------------------------------- metric configuration (BASELINE.json configs[1])
FULL_UNET = dict(motion_mask=True, motion_strength=True)        # the v1.02 architecture: every other ctor default


def fullsize_inputs(frames=16, lat=64, seed=1234):
    """One CFG-doubled UNet call of the benchmarked step (same construction as bench.py `synthetic_inputs` +
    `LatentToVideoPipeline.denoise`): sample [2,4,frames,lat,lat], cond [2,4,1,lat,lat], text [2,77,1024] = [neg; text]."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    mask = torch.zeros(1, 1, 1, lat, lat)
    mask[..., lat // 4: lat - lat // 4, lat // 4: lat - lat // 4] = 1
    latents, cond, text, neg = r(1, 4, frames, lat, lat), r(1, 4, 1, lat, lat), r(1, 77, 1024), r(1, 77, 1024)
    return dict(sample=torch.cat([latents, latents]), cond=torch.cat([cond, cond]), mask=mask,
                text=torch.cat([neg, text]), motion=torch.tensor([3.0]), t=951)


def fullsize_oracle():
    """The full architecture with seeded weights (fp32, CPU); returns (module, state dict)."""
    import oracle
    torch.manual_seed(0)
    ref = oracle.UNet3DConditionModel(**FULL_UNET).eval()
    state = seeded_state(ref)
    ref.load_state_dict(state)
    return ref, state


# ---------------------------------------------------------------------------- Stable-Video-Diffusion path (BASELINE.json configs[3])
SMALL_SVD_UNET = dict(in_channels=9, block_out_channels=(64, 128, 256, 256), num_attention_heads=(1, 2, 4, 4),
                      cross_attention_dim=128, addition_time_embed_dim=64, projection_class_embeddings_input_dim=192,
                      num_frames=4)
SMALL_SVD_VAE = dict(block_out_channels=(32, 64, 128, 128))
FULL_SVD_UNET = dict(in_channels=9, num_frames=14)       # stable-video-diffusion-img2vid + the reference's mask channel


def svd_state(ref, seed=0):
    """Seeded weights; the AlphaBlender mix factors are drawn at random so every blend is a real mixture."""
    state = seeded_state(ref, seed)
    g = torch.Generator().manual_seed(seed + 5)
    for k in state:
        if k.endswith("mix_factor"):
            state[k] = torch.randn(1, generator=g)
    return state


def svd_unet_inputs(b=2, frames=14, h=72, w=128, channels=9, text_len=1, text_dim=1024, seed=4321):
    """One CFG-doubled SVD UNet call as MaskStableVideoDiffusionPipeline assembles it (models/pipeline.py:417-431):
    [mask | scaled noisy latents | image latents] per frame, the unconditional half with zero image latents / embedding."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    lat = r(1, frames, 4, h, w)
    cond = r(1, 1, 4, h, w).repeat(1, frames, 1, 1, 1)
    emb = r(1, text_len, text_dim)
    mask = torch.zeros(1, frames, 1, h, w)
    mask[..., h // 4: h - h // 4, w // 4: w - w // 4] = 1
    parts = [mask.repeat(b, 1, 1, 1, 1)] if channels == 9 else []
    x = torch.cat(parts + [lat.repeat(b, 1, 1, 1, 1), torch.cat([torch.zeros_like(cond), cond])[:b]], dim=2)
    ctx = torch.cat([torch.zeros_like(emb), emb])[:b]
    ids = torch.tensor([[6.0, 127.0, 0.02]]).repeat(b, 1)
    return dict(sample=x, t=1.3, text=ctx, ids=ids)


def fullsize_svd_oracle():
    """The full SVD architecture with seeded weights (fp32, CPU); returns (module, state dict)."""
    import oracle.svd as O
    torch.manual_seed(0)
    ref = O.UNetSpatioTemporalConditionModel(**FULL_SVD_UNET).eval()
    state = svd_state(ref)
    ref.load_state_dict(state)
    return ref, state
