"""Stable-Video-Diffusion path (reference models/pipeline.py:223-731, train_svd.py:726-826) on the
CPU: the product modules driven through the SIMT emulator against the fp32 oracle (oracle/svd.py), the new glue kernels
against torch expressions, the Euler schedule against the oracle's independent restatement."""
import numpy as np
import pytest
import torch

import oracle.svd as O
from animate_anything_amd import ops
from animate_anything_amd._lib import AA_ACT_SILU
from animate_anything_amd.schedulers import EulerDiscreteScheduler
from animate_anything_amd.svd_pipeline import (MaskStableVideoDiffusionPipeline, TextStableVideoDiffusionPipeline,
                                               _resize_with_antialiasing)
from animate_anything_amd.svd_unet import TransformerSpatioTemporalModel, UNetSpatioTemporalConditionModel
from animate_anything_amd.svd_vae import AutoencoderKLTemporalDecoder
from util import rel_err, svd_state

TINY_SVD_UNET = dict(in_channels=9, block_out_channels=(64, 128),
                     down_block_types=("CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal"),
                     up_block_types=("UpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal"), layers_per_block=1,
                     cross_attention_dim=64, num_attention_heads=(1, 2), addition_time_embed_dim=32,
                     projection_class_embeddings_input_dim=96, num_frames=3)
TINY_SVD_VAE = dict(block_out_channels=(32, 64), layers_per_block=1)


def tiny_unet(in_channels=9):
    torch.manual_seed(0)
    cfg = dict(TINY_SVD_UNET, in_channels=in_channels)
    ref = O.UNetSpatioTemporalConditionModel(**cfg).eval()
    state = svd_state(ref)
    ref.load_state_dict(state)
    net = UNetSpatioTemporalConditionModel(**cfg).eval()
    assert set(net.state_dict().keys()) == set(state.keys())
    net.load_state_dict(state)
    return ref, net.half()


def tiny_vae():
    torch.manual_seed(1)
    ref = O.AutoencoderKLTemporalDecoder(**TINY_SVD_VAE).eval()
    state = svd_state(ref, 1)
    ref.load_state_dict(state)
    net = AutoencoderKLTemporalDecoder(**TINY_SVD_VAE).eval()
    assert set(net.state_dict().keys()) == set(state.keys())
    net.load_state_dict(state)
    return ref, net.half()


# ------------------------------------------------------------------------------------------- glue kernels
def test_blend_kernel(emu):
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(60, 24, generator=g).half(), torch.randn(60, 24, generator=g).half()
    rv = torch.randn(5, 40, generator=g).half()[:, 8:32]               # a column slice of a wider matrix
    got = ops.blend(x, y, 0.3, 0.7)
    assert rel_err(got, 0.3 * x.float() + 0.7 * y.float()) < 2e-3
    got = ops.blend(x, rowvec=rv, rowvec_div=4, rowvec_mod=5)
    idx = (torch.arange(60) // 4) % 5
    assert rel_err(got, x.float() + rv.float()[idx]) < 2e-3
    got = ops.blend(x, y, act=AA_ACT_SILU)
    assert rel_err(got, torch.nn.functional.silu(x.float() + y.float())) < 2e-3
    got = ops.blend(x, rowvec=rv, rowvec_div=1, rowvec_mod=5, out=x.clone())
    assert rel_err(got, x.float() + rv.float()[torch.arange(60) % 5]) < 2e-3


@pytest.mark.parametrize("with_mask,dtype", [(True, torch.float16), (False, torch.bfloat16)])
def test_pack_frames_kernel(emu, with_mask, dtype):
    g = torch.Generator().manual_seed(1)
    b, f, h, w = 2, 3, 4, 5
    lat = torch.randn(1, f, 4, h, w, generator=g)                        # fp32, shared by both guidance halves
    cond = torch.randn(b, f, 4, h, w, generator=g).to(dtype)
    mask = (torch.rand(2, f, 1, h, w, generator=g) > 0.5).to(dtype) if with_mask else None
    scale = torch.tensor([0.37])
    oc = 16 if with_mask else 8
    got = ops.pack_frames([mask, lat, cond], b, dtype, scale, 1, oc)
    want = torch.cat(([mask.float().expand(b, -1, -1, -1, -1)] if with_mask else []) +
                     [(lat * 0.37).expand(b, -1, -1, -1, -1), cond.float()], dim=2)            # [b, f, C, h, w]
    c = want.shape[2]
    want = want.permute(0, 1, 3, 4, 2).reshape(-1, c)
    assert got.shape == (b * f * h * w, oc)
    assert rel_err(got[:, :c], want) < (2e-3 if dtype == torch.float16 else 1e-2)
    assert c == oc or got[:, c:].abs().max() == 0


@pytest.mark.parametrize("guided", [True, False])
def test_cfg_euler_step_kernel(emu, guided):
    g = torch.Generator().manual_seed(2)
    clips, f, c, h, w = 2, 3, 4, 3, 5
    v = torch.randn((2 if guided else 1) * clips * f * h * w, 4, generator=g).half()
    x = torch.randn(clips, f, c, h, w, generator=g)
    scale_f = torch.linspace(1.0, 3.0, f)
    v5 = v.float().reshape(-1, clips, f, h, w, c).permute(0, 1, 2, 5, 3, 4)
    vv = v5[0] + scale_f.reshape(1, f, 1, 1, 1) * (v5[1] - v5[0]) if guided else v5[0]
    want = 0.9 * x + (-0.2) * vv
    nt, ns = torch.zeros(4), torch.zeros(1)
    ops.cfg_euler_step_tokens(v, x, scale_f if guided else None, 0.9, -0.2, next_t=nt, next_t_value=1.25, next_scale=ns,
                              next_scale_value=0.5)
    assert rel_err(x, want) < 1e-5
    assert torch.all(nt == 1.25) and ns.item() == 0.5


@pytest.mark.parametrize("shape", ["dma", "generic", "splitk"])
def test_conv_gemm_acc_scale(emu, shape):
    """out = acc_scale * (x W^T + b) + residual: the temporal-branch blend weight of SpatioTemporalResBlock."""
    g = torch.Generator().manual_seed(3)
    m, k, n = (300, 64, 64) if shape != "splitk" else (40, 2048, 64)
    if shape == "generic":
        k = 24
    x, wt = torch.randn(m, k, generator=g).half(), (torch.randn(n, k, generator=g) / k ** 0.5).half()
    bias, res = torch.randn(n, generator=g).half(), torch.randn(m, n, generator=g).half()
    pw = ops.pack_weight(wt, bias)
    old = ops.K_SPLITS
    ops.K_SPLITS = 4 if shape == "splitk" else 0
    try:
        got = ops.conv_gemm(x, pw, ops.linear_geom(m), residual=res, acc_scale=0.375)
    finally:
        ops.K_SPLITS = old
    want = 0.375 * (x.float() @ wt.float().t() + bias.float()) + res.float()
    assert rel_err(got, want) < 3e-3


def test_attention_kv_table_addressing(emu):
    """K / V operand addressed as a table: sequence n = clip*pixels + pixel reads context n % clips (diffusers 0.24.0)."""
    g = torch.Generator().manual_seed(4)
    clips, frames, hw, heads, L = 2, 3, 5, 2, 4
    q = torch.randn(clips * frames * hw, heads * 64, generator=g).half()
    kv = torch.randn(clips * L, 2 * heads * 64, generator=g).half()
    got = ops.attention(q, 0, kv, 0, kv, heads * 64, heads, clips, hw, frames, L, (frames * hw, 1, hw), (L, 0, 1),
                        kv_seq_mod=clips)
    qf = q.float().reshape(clips, frames, hw, heads, 64)
    kf = kv.float().reshape(clips, L, 2, heads, 64)
    want = torch.zeros_like(qf)
    for b in range(clips):
        for p in range(hw):
            c = (b * hw + p) % clips
            for hd in range(heads):
                s = qf[b, :, p, hd] @ kf[c, :, 0, hd].t() / 8.0
                want[b, :, p, hd] = s.softmax(-1) @ kf[c, :, 1, hd]
    assert rel_err(got, want.reshape(-1, heads * 64)) < 3e-3


# ------------------------------------------------------------------------------------------- scheduler
def test_euler_schedule_matches_oracle():
    p, o = EulerDiscreteScheduler(), O.EulerDiscreteScheduler()
    p.set_timesteps(25)
    o.set_timesteps(25)
    assert abs(p.init_noise_sigma - o.init_noise_sigma) < 1e-4 and abs(p.init_noise_sigma - (700.0 ** 2 + 1) ** 0.5) < 1e-3
    assert torch.allclose(p.timesteps, o.timesteps, atol=1e-6)
    assert abs(float(p.timesteps[0]) - 0.25 * np.log(700.0)) < 1e-6
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, 4, 5, 5, generator=g) * p.init_noise_sigma
    xo = x.clone()
    for i, t in enumerate(p.timesteps):
        v = torch.randn(x.shape, generator=g)
        assert torch.allclose(p.scale_model_input(x, t), o.scale_model_input(xo, t), rtol=1e-5, atol=1e-6)
        k = p.coefficients(i)
        fused = k["c_x"] * x + k["c_v"] * v
        x = p.step(v, t, x).prev_sample
        xo = o.step(v, t, xo).prev_sample
        assert rel_err(x, xo) < 1e-5 and rel_err(fused, xo) < 1e-5
    assert p._sig[-1] == 0.0


def test_resize_with_antialiasing_matches_oracle():
    x = torch.rand(1, 3, 40, 72, generator=torch.Generator().manual_seed(0)) * 2 - 1
    assert torch.allclose(_resize_with_antialiasing(x, (16, 16)), O.resize_with_antialiasing(x, (16, 16)), atol=1e-6)


# ------------------------------------------------------------------------------------------- modules on the emulator
@pytest.mark.parametrize("text_len,pixel_major", [(1, True), (5, True), (1, False), (3, False)])
def test_tiny_svd_unet_matches_oracle(emu, text_len, pixel_major, monkeypatch):
    """The context is one token (CLIP image embedding: row-vector fast paths) or several (text conditioning: attention
    kernels); `pixel_major` = the diffusers==0.24.0 ordering of the temporal blocks' context (what the reference runs)."""
    ref, net = tiny_unet()
    monkeypatch.setattr(TransformerSpatioTemporalModel, "pixel_major_time_context", pixel_major)
    monkeypatch.setattr(O.TransformerSpatioTemporalModel, "pixel_major_time_context", pixel_major, raising=False)
    g = torch.Generator().manual_seed(7)
    b, f, h, w = 2, 3, 4, 6
    x, ctx = torch.randn(b, f, 9, h, w, generator=g), torch.randn(b, text_len, 64, generator=g)
    ids = torch.tensor([[6.0, 127.0, 0.02]]).repeat(b, 1)
    with torch.no_grad():
        want = ref(x, 1.2, ctx, ids).sample
        got = net(x.half(), 1.2, ctx.half(), ids).sample
    assert got.shape == want.shape == (b, f, 4, h, w)
    assert rel_err(got, want) < 1e-2


@pytest.mark.parametrize("seed", [1, 2])
def test_tiny_svd_unet_random_tile_assignments(emu, seed):
    """Tile fuzzing on the emulator (ops.TILE_PICKER): every contraction of the tiny SVD UNet - spatio-temporal resnets with the
    `acc_scale` blend epilogue, both transformer kinds, the row-vector context path - on a random eligible tile of the table."""
    import random
    ref, net = tiny_unet()
    g = torch.Generator().manual_seed(7)
    b, f, h, w = 2, 3, 4, 6
    x, ctx = torch.randn(b, f, 9, h, w, generator=g), torch.randn(b, 1, 64, generator=g)
    ids = torch.tensor([[6.0, 127.0, 0.02]]).repeat(b, 1)
    rng, used = random.Random(seed), {}

    def pick(key, cands):
        if key not in used:
            used[key] = rng.choice(cands)
        return used[key]
    ops.TILE_PICKER = pick
    try:
        with torch.no_grad():
            got = net(x.half(), 1.2, ctx.half(), ids).sample
    finally:
        ops.TILE_PICKER = None
    with torch.no_grad():
        want = ref(x, 1.2, ctx, ids).sample
    assert len({c[0] for c in used.values()}) >= 8, used
    assert rel_err(got, want) < 1e-2, (rel_err(got, want), sorted(used.items(), key=str))


def test_tiny_svd_unet_batch3_and_8_channels(emu):
    """h*w not a multiple of the batch (the general context-table attention instead of the row-vector blend) and the
    8-input-channel (no mask) model of the plain SVD checkpoint."""
    ref, net = tiny_unet(in_channels=8)
    g = torch.Generator().manual_seed(8)
    b, f, h, w = 3, 2, 4, 4
    x, ctx = torch.randn(b, f, 8, h, w, generator=g), torch.randn(b, 1, 64, generator=g)
    ids = torch.tensor([[6.0, 127.0, 0.02]]).repeat(b, 1)
    with torch.no_grad():
        want = ref(x, torch.tensor([0.3] * b), ctx, ids).sample
        got = net(x.half(), torch.tensor([0.3] * b), ctx.half(), ids).sample
    assert rel_err(got, want) < 1e-2


def test_tiny_temporal_vae_matches_oracle(emu):
    ref, net = tiny_vae()
    g = torch.Generator().manual_seed(9)
    z, img = torch.randn(6, 4, 5, 6, generator=g), torch.randn(1, 3, 16, 24, generator=g)
    with torch.no_grad():
        want, got = ref.decode(z, num_frames=3).sample, net.decode(z.half(), num_frames=3).sample
        wenc, genc = ref.encode(img).latent_dist.mode(), net.encode(img.half()).latent_dist.mode()
    assert got.shape == want.shape == (6, 3, 10, 12)
    assert rel_err(got, want) < 1e-2 and rel_err(genc, wenc) < 1e-2
    with pytest.raises(ValueError):
        net.decode(z.half(), num_frames=4)


# ------------------------------------------------------------------------------------------- pipelines on the emulator
def _pipeline_case(seed=11, b=1, f=2, H=16, W=24):
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(b, 3, H, W, generator=g) * 2 - 1
    emb = torch.randn(b, 1, 64, generator=g)
    latents = torch.randn(b, f, 4, H // 2, W // 2, generator=g)        # the tiny VAE scales by 2
    mask = torch.zeros(1, H // 2, W // 2)
    mask[:, 2:6, 3:9] = 1
    return image, emb, latents, mask


@pytest.mark.parametrize("fused", [True, False])
def test_mask_svd_pipeline_matches_oracle(emu, fused):
    """reference models/pipeline.py:223-466 end to end (tiny models, 2 Euler steps, per-frame guidance 1..3): the fused loop
    (UNet session + aa_cfg_euler_step_tokens) and the generic loop (module forward + scheduler.step)."""
    ref_u, net_u = tiny_unet()
    ref_v, net_v = tiny_vae()
    image, emb, latents, mask = _pipeline_case()
    seed = 123
    noise = torch.randn(image.shape, generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        want = O.svd_pipeline(ref_u, ref_v, O.EulerDiscreteScheduler(), image, torch.cat([torch.zeros_like(emb), emb]), mask=mask,
                              num_frames=2, num_inference_steps=2, latents=latents.clone(), aug_noise=noise,
                              output_type="latent")
        want_frames = O.decode_latents(ref_v, want, 2, 2)
        pipe = MaskStableVideoDiffusionPipeline(net_v, None, net_u, EulerDiscreteScheduler())
        pipe.fused_step = fused
        got = pipe(image, height=16, width=24, num_frames=2, num_inference_steps=2, latents=latents.clone(), mask=mask,
                   generator=torch.Generator().manual_seed(seed), output_type="latent", image_embeddings=emb).frames
        got_frames = pipe.decode_latents(got, 2, 2)
    assert got.shape == want.shape == (1, 2, 4, 8, 12)
    assert rel_err(got, want) < 2e-2
    assert got_frames.shape == want_frames.shape == (1, 3, 2, 16, 24)
    assert rel_err(got_frames, want_frames) < 3e-2
    mse = ((got.float() - want) ** 2).mean().item() / (want ** 2).mean().item()
    assert mse < 1e-3


def test_text_svd_pipeline_matches_oracle(emu):
    """reference models/pipeline.py:468-731 with condition_type='text' (5-token context), an 8-channel UNet (no motion mask)
    and a caller-supplied condition latent."""
    ref_u, net_u = tiny_unet(in_channels=8)
    ref_v, net_v = tiny_vae()
    image, _, latents, _ = _pipeline_case(seed=12)
    g = torch.Generator().manual_seed(13)
    pe, ne = torch.randn(1, 5, 64, generator=g), torch.randn(1, 5, 64, generator=g)
    cond = torch.randn(1, 2, 4, 8, 12, generator=g)
    with torch.no_grad():
        want = O.svd_pipeline(ref_u, ref_v, O.EulerDiscreteScheduler(), image, torch.cat([ne, pe]), num_frames=2,
                              num_inference_steps=2, latents=latents.clone(), condition_latent=cond, output_type="latent",
                              min_guidance_scale=1.5, max_guidance_scale=2.5)
        pipe = TextStableVideoDiffusionPipeline(net_v, None, net_u, EulerDiscreteScheduler())
        got = pipe(image, prompt_embeds=pe.half(), negative_prompt_embeds=ne.half(), height=16, width=24, num_frames=2,
                   num_inference_steps=2, latents=latents.clone(), condition_type="text", condition_latent=cond.half(),
                   min_guidance_scale=1.5, max_guidance_scale=2.5, output_type="latent", return_dict=False)
    assert rel_err(got, want) < 2e-2


def test_svd_pipeline_np_output_and_errors(emu):
    ref_u, net_u = tiny_unet()
    ref_v, net_v = tiny_vae()
    image, emb, latents, mask = _pipeline_case()
    pipe = MaskStableVideoDiffusionPipeline(net_v, None, net_u, EulerDiscreteScheduler())
    with torch.no_grad():
        out = pipe(image, height=16, width=24, num_frames=2, num_inference_steps=1, latents=latents, mask=mask,
                   output_type="np", image_embeddings=emb, decode_chunk_size=1)
    assert out.frames.shape == (1, 2, 16, 24, 3) and out.frames.min() >= 0 and out.frames.max() <= 1
    with pytest.raises(ValueError):
        pipe(image, height=30, width=24, num_frames=2, image_embeddings=emb, mask=mask)          # not divisible by 8
    with pytest.raises(ValueError):
        pipe(image, height=16, width=24, num_frames=2, mask=mask)                                # no image encoder, no embeddings


def test_svd_unet_refuses_cpu_tensors():
    _, net = tiny_unet()
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 2, 9, 4, 4).half(), 1.0, torch.zeros(1, 1, 64).half(), torch.zeros(1, 3))


# ------------------------------------------------------------------------------------------- checkpoints (no kernels involved)
def test_svd_pipeline_save_load_roundtrip_and_convert_svd(emu, tmp_path):
    """save_pretrained / from_pretrained of the whole pipeline (diffusers directory layout) and the reference's `convert_svd`
    (train_svd.py:93-103): the 9-channel UNet it builds ignores the new mask channel (zero weights) and otherwise IS the
    8-channel model."""
    from animate_anything_amd import eval_svd
    from animate_anything_amd.svd_pipeline import StableVideoDiffusionPipeline
    _, net8 = tiny_unet(in_channels=8)
    _, vae = tiny_vae()
    pipe = StableVideoDiffusionPipeline(vae.float(), None, net8.float(), EulerDiscreteScheduler())
    pipe.save_pretrained(str(tmp_path / "svd8"))
    again = StableVideoDiffusionPipeline.from_pretrained(str(tmp_path / "svd8"))
    assert again.unet.config.in_channels == 8 and again.scheduler.config.sigma_max == 700.0
    assert all(torch.equal(a, b) for a, b in zip(pipe.unet.state_dict().values(), again.unet.state_dict().values()))
    assert all(torch.equal(a, b) for a, b in zip(pipe.vae.state_dict().values(), again.vae.state_dict().values()))
    new = eval_svd.convert_svd(str(tmp_path / "svd8"), str(tmp_path / "svd9"))
    net9 = StableVideoDiffusionPipeline.from_pretrained(str(tmp_path / "svd9")).unet
    assert net9.config.in_channels == 9 and net9.conv_in.weight[:, 0].abs().max() == 0
    assert torch.equal(net9.conv_in.weight[:, 1:], again.unet.conv_in.weight) and new.unet.config.in_channels == 9
    g = torch.Generator().manual_seed(21)
    b, f, h, w = 2, 2, 4, 6
    x8, ctx = torch.randn(b, f, 8, h, w, generator=g), torch.randn(b, 1, 64, generator=g)
    x9 = torch.cat([torch.rand(b, f, 1, h, w, generator=g), x8], dim=2)
    ids = torch.tensor([[6.0, 127.0, 0.02]]).repeat(b, 1)
    with torch.no_grad():
        y8 = again.unet.half()(x8.half(), 0.7, ctx.half(), ids).sample
        y9 = net9.half()(x9.half(), 0.7, ctx.half(), ids).sample
    assert rel_err(y9, y8) < 2e-3


def test_encode_image_through_a_transformers_clip_vision_tower():
    """`_encode_image` (diffusers StableVideoDiffusionPipeline): PIL image -> antialiased 224x224 -> CLIP normalisation -> the
    `transformers` CLIPVisionModelWithProjection the reference loads (a tiny random-init one here) -> [uncond zeros; embedding]."""
    from PIL import Image
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.manual_seed(0)
    cfg = CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, image_size=224,
                           patch_size=32, projection_dim=64)
    tower = CLIPVisionModelWithProjection(cfg).eval()
    _, net = tiny_unet()
    pipe = MaskStableVideoDiffusionPipeline(None, tower, net, EulerDiscreteScheduler())
    img = Image.fromarray((np.random.default_rng(0).random((90, 160, 3)) * 255).astype(np.uint8))
    with torch.no_grad():
        e = pipe._encode_image(img, torch.device("cpu"), 1, True)
        x = pipe.image_processor.preprocess(img)
        x = (_resize_with_antialiasing(x, (224, 224)) + 1.0) / 2.0
        mean = torch.tensor(pipe._CLIP_MEAN).reshape(1, 3, 1, 1)
        std = torch.tensor(pipe._CLIP_STD).reshape(1, 3, 1, 1)
        want = tower((x - mean) / std).image_embeds
    assert e.shape == (2, 1, 64) and e[0].abs().max() == 0
    assert torch.allclose(e[1, 0], want[0], atol=1e-6)
    ready = torch.randn(1, 1, 64)
    assert torch.equal(pipe._encode_image(img, torch.device("cpu"), 1, False, ready), ready)


def test_parameter_counts_of_the_real_architectures():
    """Structural known-answer test: with the default (stable-video-diffusion-img2vid) configuration the modules have
    1 524 623 082 and 97 742 847 parameters - the sizes of the published fp16 checkpoints (3.05 GB UNet, 196 MB VAE = 2 bytes per
    parameter) - and the reference's mask channel adds 320 x 3 x 3 input weights."""
    with torch.device("meta"):
        unet8 = UNetSpatioTemporalConditionModel()
        unet9 = UNetSpatioTemporalConditionModel(in_channels=9)
        vae = AutoencoderKLTemporalDecoder()
    count = lambda m: sum(p.numel() for p in m.parameters())
    assert count(unet8) == 1524623082 and count(unet9) - count(unet8) == 320 * 9
    assert count(vae) == 97742847


def test_variant_checkpoints_and_callback_latents(tmp_path):
    """`from_pretrained(..., variant="fp16")` finds `diffusion_pytorch_model.fp16.safetensors` (the reference's SVD
    eval loads that way, train_svd.py:806-811), and the generic denoising loop takes `latents` back from `callback_on_step_end`
    (models/pipeline.py:445-447) and accepts a batch-2 mask when guidance is off."""
    import os
    from animate_anything_amd._ckpt import load_state
    from util import SMALL_SVD_VAE
    vae = AutoencoderKLTemporalDecoder(**SMALL_SVD_VAE)
    d = tmp_path / "vae"
    vae.save_pretrained(str(d))
    os.rename(d / "diffusion_pytorch_model.safetensors", d / "diffusion_pytorch_model.fp16.safetensors")
    with pytest.raises(FileNotFoundError):
        AutoencoderKLTemporalDecoder.from_pretrained(str(d))
    again = AutoencoderKLTemporalDecoder.from_pretrained(str(d), variant="fp16")
    for k, v in vae.state_dict().items():
        assert torch.equal(v, again.state_dict()[k])
    assert set(load_state(str(d), "diffusion_pytorch_model", "fp16").keys()) == set(vae.state_dict().keys())


def test_vae_overflow_retry_leaves_the_fp16_weights_untouched(emu, monkeypatch):
    """the bf16 retry of `_encode_vae_image` (the reference upcasts to fp32 under `force_upcast`, models/pipeline.py:373-383)
    runs on a copy of the encoder - the live fp16 VAE parameters are bit-identical afterwards and the result is a finite latent."""
    _, net_v = tiny_vae()
    before = {k: v.clone() for k, v in net_v.state_dict().items()}
    pipe = MaskStableVideoDiffusionPipeline(net_v, None, None, EulerDiscreteScheduler())
    image = torch.rand(1, 3, 8, 8, generator=torch.Generator().manual_seed(3)) * 2 - 1
    cls = type(net_v)
    real, calls = cls.encode, []

    def encode(self, x):
        out = real(self, x)
        calls.append(x.dtype)
        if len(calls) == 1:                                      # the fp16 pass "overflows"
            out.latent_dist.mean[...] = float("inf")
        return out
    monkeypatch.setattr(cls, "encode", encode)
    with torch.no_grad():
        lat = pipe._encode_vae_image(image.half(), torch.device("cpu"), 1, True)
    assert calls == [torch.float16, torch.bfloat16]
    assert lat.dtype == torch.float16 and torch.isfinite(lat).all() and lat.shape[0] == 2 and (lat[0] == 0).all()
    after = net_v.state_dict()
    assert all(after[k].dtype == before[k].dtype and torch.equal(after[k], before[k]) for k in before)
    monkeypatch.setattr(cls, "encode", real)
    with torch.no_grad():
        want = net_v.encode(image.half()).latent_dist.mode()
    assert rel_err(lat[1:], want) < 3e-2                         # bf16 storage against fp16 storage
