"""Stable-Video-Diffusion path on the GPU (BASELINE.json configs[3]): the HIP modules against
the fp32 CPU oracle (oracle/svd.py) - small and full architecture, the committed golden output at 14 x 72x128 latents
(576x1024 pixels), the temporal-decoder VAE, both reference pipelines and the `train_svd.py --eval` driver."""
import json
import os

import numpy as np
import pytest
import torch
import yaml
from PIL import Image

import oracle.svd as O
from animate_anything_amd import ops
from animate_anything_amd.schedulers import EulerDiscreteScheduler
from animate_anything_amd.svd_pipeline import MaskStableVideoDiffusionPipeline, TextStableVideoDiffusionPipeline
from animate_anything_amd.svd_unet import TransformerSpatioTemporalModel, UNetSpatioTemporalConditionModel
from animate_anything_amd.svd_vae import AutoencoderKLTemporalDecoder
from util import (FULL_SVD_UNET, SMALL_SVD_UNET, SMALL_SVD_VAE, fullsize_svd_oracle, rel_err, svd_state, svd_unet_inputs)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "svd_unet_fullsize_14x72x128.pt")


def _pair(ref_cls, net_cls, cfg, dtype, seed=0):
    torch.manual_seed(seed)
    ref = ref_cls(**cfg).eval()
    state = svd_state(ref, seed)
    ref.load_state_dict(state)
    net = net_cls(**cfg).eval()
    net.load_state_dict(state)
    return ref, net.to("cuda", dtype)


def _run(net, i, dt):
    return net(i["sample"].to("cuda", dt), i["t"], i["text"].to("cuda", dt), i["ids"].cuda()).sample


@pytest.mark.parametrize("dtype,text_len,pixel_major", [(torch.float16, 1, True), (torch.bfloat16, 1, True),
                                                        (torch.float16, 5, True), (torch.float16, 1, False), (torch.float16, 77, False)])
def test_small_svd_unet_matches_oracle(dtype, text_len, pixel_major, monkeypatch):
    monkeypatch.setattr(TransformerSpatioTemporalModel, "pixel_major_time_context", pixel_major)
    monkeypatch.setattr(O.TransformerSpatioTemporalModel, "pixel_major_time_context", pixel_major)
    ref, net = _pair(O.UNetSpatioTemporalConditionModel, UNetSpatioTemporalConditionModel, SMALL_SVD_UNET, dtype)
    i = svd_unet_inputs(2, 4, 16, 24, text_len=text_len, text_dim=128)
    with torch.no_grad():
        want = ref(i["sample"], i["t"], i["text"], i["ids"]).sample
        got = _run(net, i, dtype)
        net.enable_graph()
        replay = [_run(net, i, dtype).clone() for _ in range(2)]
    assert got.shape == want.shape == (2, 4, 4, 16, 24)
    assert rel_err(got, want) < (2e-2 if dtype == torch.float16 else 8e-2)
    assert torch.equal(replay[0], got) and torch.equal(replay[1], got)          # hipGraph replay == eager, bit for bit


def test_full_architecture_svd_unet_matches_oracle():
    """The real 1.5 G-parameter architecture (320/640/1280/1280, heads 5/10/20/20, 9 input channels) on a small grid: every
    layer type at its real width, oracle live on the host."""
    ref, state = fullsize_svd_oracle()
    net = UNetSpatioTemporalConditionModel(**FULL_SVD_UNET).eval()
    net.load_state_dict(state)
    net = net.to("cuda", torch.float16)
    i = svd_unet_inputs(2, 3, 16, 24)
    with torch.no_grad():
        want = ref(i["sample"], i["t"], i["text"], i["ids"]).sample
        got = _run(net, i, torch.float16)
    assert rel_err(got, want) < 2e-2
    assert ((got.float().cpu() - want) ** 2).mean().item() < 1e-3 * max((want ** 2).mean().item(), 1.0)


@pytest.mark.skipif(not os.path.exists(GOLDEN), reason="tests/golden/make_svd_golden.py has not been run")
def test_fullsize_svd_unet_matches_golden():
    """BASELINE.json configs[3]: 14 frames x 72x128 latents, CFG batch 2 = 258048 tokens at the first level."""
    gold = torch.load(GOLDEN)
    _, state = fullsize_svd_oracle()
    net = UNetSpatioTemporalConditionModel(**FULL_SVD_UNET).eval()
    net.load_state_dict(state)
    del state
    net = net.to("cuda", torch.float16)
    net.enable_graph()
    i = svd_unet_inputs(2, 14, 72, 128)
    with torch.no_grad():
        got = _run(net, i, torch.float16).float().cpu()
    want = gold["out"].float()
    assert got.shape == want.shape == (2, 14, 4, 72, 128)
    assert rel_err(got, want) < 3e-2
    assert ((got - want) ** 2).mean().item() < 1e-3 * max((want ** 2).mean().item(), 1.0)


def test_small_temporal_vae_matches_oracle():
    ref, net = _pair(O.AutoencoderKLTemporalDecoder, AutoencoderKLTemporalDecoder, SMALL_SVD_VAE, torch.float16, seed=1)
    g = torch.Generator().manual_seed(3)
    z, img = torch.randn(8, 4, 9, 12, generator=g), torch.randn(2, 3, 72, 96, generator=g)
    with torch.no_grad():
        want, got = ref.decode(z, num_frames=4).sample, net.decode(z.cuda().half(), num_frames=4).sample
        wenc, genc = ref.encode(img).latent_dist.mode(), net.encode(img.cuda().half()).latent_dist.mode()
    assert got.shape == want.shape == (8, 3, 72, 96)
    assert rel_err(got, want) < 2e-2 and rel_err(genc, wenc) < 2e-2


def test_fullsize_temporal_vae_matches_oracle_on_the_gpu():
    """The real VAE widths (128/256/512/512): decode of a 3-frame chunk and encode of one image at 288x512 against the SAME oracle
    modules evaluated in fp32 by torch on the GPU (the CPU oracle would need minutes per frame; at 576x1024 the fp32 torch
    convolutions alone take 2.5 minutes), then the whole 14-frame decode at 576x1024 (BASELINE configs[3]) in one call (2.1 GB per
    128-channel activation): finite, right shape."""
    torch.manual_seed(2)
    ref = O.AutoencoderKLTemporalDecoder().eval()
    state = svd_state(ref, 2)
    ref.load_state_dict(state)
    net = AutoencoderKLTemporalDecoder().eval()
    net.load_state_dict(state)
    net = net.to("cuda", torch.float16)
    ref = ref.to("cuda")
    g = torch.Generator().manual_seed(5)
    z = torch.randn(14, 4, 72, 128, generator=g).cuda()
    img = (torch.rand(1, 3, 288, 512, generator=g) * 2 - 1).cuda()
    zs = z[:3, :, :36, :64].contiguous()
    with torch.no_grad():
        want = ref.decode(zs, num_frames=3).sample
        got = net.decode(zs.half(), num_frames=3).sample
        wenc = ref.encode(img).latent_dist.mode()
        genc = net.encode(img.half()).latent_dist.mode()
        full = net.decode(z.half(), num_frames=14).sample
    assert got.shape == want.shape == (3, 3, 288, 512)
    assert rel_err(got, want) < 3e-2 and rel_err(genc, wenc) < 3e-2
    assert full.shape == (14, 3, 576, 1024) and torch.isfinite(full).all()


@pytest.mark.parametrize("graph", [True, False])
def test_mask_svd_pipeline_matches_oracle(graph):
    """reference models/pipeline.py:223-466 end to end: small models, 6 Euler steps, per-frame guidance 1..3, motion mask."""
    ref_u, net_u = _pair(O.UNetSpatioTemporalConditionModel, UNetSpatioTemporalConditionModel, SMALL_SVD_UNET, torch.float16)
    ref_v, net_v = _pair(O.AutoencoderKLTemporalDecoder, AutoencoderKLTemporalDecoder, SMALL_SVD_VAE, torch.float16, seed=1)
    g = torch.Generator().manual_seed(11)
    H, W, f = 128, 192, 4
    image = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    emb = torch.randn(1, 1, 128, generator=g)
    latents = torch.randn(1, f, 4, H // 8, W // 8, generator=g)
    mask = torch.zeros(1, H // 8, W // 8)
    mask[:, 4:12, 6:18] = 1
    noise = torch.randn(image.shape, generator=torch.Generator(device="cuda").manual_seed(5), device="cuda").cpu()
    steps = 6
    with torch.no_grad():
        want = O.svd_pipeline(ref_u, ref_v, O.EulerDiscreteScheduler(), image, torch.cat([torch.zeros_like(emb), emb]), mask=mask,
                              num_frames=f, num_inference_steps=steps, latents=latents.clone(), aug_noise=noise, output_type="latent")
        want_frames = O.decode_latents(ref_v, want, f, 2)
        pipe = MaskStableVideoDiffusionPipeline(net_v, None, net_u, EulerDiscreteScheduler())
        if graph:
            net_u.enable_graph()
        got = pipe(image.cuda(), height=H, width=W, num_frames=f, num_inference_steps=steps, latents=latents.cuda(), mask=mask.cuda(),
                   generator=torch.Generator(device="cuda").manual_seed(5), output_type="latent", image_embeddings=emb.cuda()).frames
        got_frames = pipe.decode_latents(got, f, 2)
        pipe.fused_step = False                       # the reference's own per-step sequence through the module forward
        generic = pipe(image.cuda(), height=H, width=W, num_frames=f, num_inference_steps=steps, latents=latents.cuda(),
                       mask=mask.cuda(), generator=torch.Generator(device="cuda").manual_seed(5), output_type="latent",
                       image_embeddings=emb.cuda()).frames
    mse = ((got.float().cpu() - want) ** 2).mean().item()
    assert mse < 1e-3 * max((want ** 2).mean().item(), 1.0), mse
    assert rel_err(got, want) < 3e-2 and rel_err(generic, want) < 3e-2
    assert rel_err(got_frames, want_frames) < 5e-2


@pytest.mark.parametrize("condition_type", ["text", "both"])
def test_text_svd_pipeline_matches_oracle(condition_type):
    """reference models/pipeline.py:468-731: condition_type='text' (77-token context) and the image + text form (:612-616,
    78 tokens), 8-channel UNet, caller-supplied condition latent."""
    cfg = dict(SMALL_SVD_UNET, in_channels=8)
    ref_u, net_u = _pair(O.UNetSpatioTemporalConditionModel, UNetSpatioTemporalConditionModel, cfg, torch.float16)
    ref_v, net_v = _pair(O.AutoencoderKLTemporalDecoder, AutoencoderKLTemporalDecoder, SMALL_SVD_VAE, torch.float16, seed=1)
    g = torch.Generator().manual_seed(13)
    H, W, f = 128, 128, 3
    image = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    pe, ne = torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
    cond = torch.randn(1, f, 4, H // 8, W // 8, generator=g)
    latents = torch.randn(1, f, 4, H // 8, W // 8, generator=g)
    clip = torch.randn(1, 1, 128, generator=g)
    ctx = torch.cat([ne, pe])
    if condition_type == "both":
        ctx = torch.cat([torch.cat([torch.zeros_like(clip), clip]), ctx], dim=1)           # [uncond; cond] x (1 + 77) tokens
    with torch.no_grad():
        want = O.svd_pipeline(ref_u, ref_v, O.EulerDiscreteScheduler(), image, ctx, num_frames=f,
                              num_inference_steps=4, latents=latents.clone(), condition_latent=cond, output_type="latent",
                              min_guidance_scale=1.5, max_guidance_scale=2.5)
        pipe = TextStableVideoDiffusionPipeline(net_v, None, net_u, EulerDiscreteScheduler())
        net_u.enable_graph()
        got = pipe(image.cuda(), prompt_embeds=pe.cuda().half(), negative_prompt_embeds=ne.cuda().half(), height=H, width=W,
                   num_frames=f, num_inference_steps=4, latents=latents.cuda(), condition_type=condition_type,
                   condition_latent=cond.cuda().half(), min_guidance_scale=1.5, max_guidance_scale=2.5, output_type="latent",
                   return_dict=False, image_embeddings=clip.cuda().half())
    assert rel_err(got, want) < 3e-2


def test_svd_glue_kernels_at_the_real_shapes():
    """aa_blend / aa_pack_frames / aa_cfg_euler_step_tokens at configs[3]'s first-level shapes against torch expressions on the GPU."""
    g = torch.Generator(device="cuda").manual_seed(0)
    b, f, h, w = 2, 14, 72, 128
    rows = b * f * h * w
    x = torch.randn(rows, 320, generator=g, device="cuda").half()
    y = torch.randn(rows, 320, generator=g, device="cuda").half()
    emb = torch.randn(f, 320, generator=g, device="cuda").half()
    assert rel_err(ops.blend(x, y, 0.3, 0.7), 0.3 * x.float() + 0.7 * y.float()) < 2e-3
    idx = (torch.arange(rows, device="cuda") // (h * w)) % f
    assert rel_err(ops.blend(x, rowvec=emb, rowvec_div=h * w, rowvec_mod=f), x.float() + emb.float()[idx]) < 2e-3
    lat = torch.randn(1, f, 4, h, w, generator=g, device="cuda")
    cond = torch.randn(b, f, 4, h, w, generator=g, device="cuda").half()
    mask = (torch.rand(2, f, 1, h, w, generator=g, device="cuda") > 0.5).half()
    scale = torch.tensor([0.0123], device="cuda")
    got = ops.pack_frames([mask, lat, cond], b, torch.float16, scale, 1, 16)
    want = torch.cat([mask.float(), (lat * 0.0123).expand(b, -1, -1, -1, -1), cond.float()], dim=2).permute(0, 1, 3, 4, 2).reshape(-1, 9)
    assert rel_err(got[:, :9], want) < 2e-3 and got[:, 9:].abs().max() == 0
    v = torch.randn(rows, 4, generator=g, device="cuda").half()
    xl = torch.randn(1, f, 4, h, w, generator=g, device="cuda")
    gs = torch.linspace(1.0, 3.0, f, device="cuda")
    v5 = v.float().reshape(2, 1, f, h, w, 4).permute(0, 1, 2, 5, 3, 4)
    want = 0.8 * xl + 0.25 * (v5[0] + gs.reshape(1, f, 1, 1, 1) * (v5[1] - v5[0]))
    ops.cfg_euler_step_tokens(v, xl, gs, 0.8, 0.25)
    assert rel_err(xl, want) < 1e-5


def test_svd_eval_driver_roundtrip(tmp_path):
    """`train_svd.py --eval` flow (train_svd.py:726-826): synthetic diffusers-layout checkpoint -> from_pretrained -> image +
    `_label.jpg` motion mask -> CLIP image embedding (vision tower from the checkpoint) -> MaskStableVideoDiffusionPipeline -> gif."""
    from animate_anything_amd import eval_svd
    torch.manual_seed(0)
    ckpt = tmp_path / "svd"
    unet = UNetSpatioTemporalConditionModel(**SMALL_SVD_UNET)
    unet.save_pretrained(str(ckpt / "unet"))
    AutoencoderKLTemporalDecoder(**SMALL_SVD_VAE).save_pretrained(str(ckpt / "vae"))
    from animate_anything_amd.clip import CLIPVisionModelWithProjection
    CLIPVisionModelWithProjection(hidden_size=128, intermediate_size=256, projection_dim=128, num_hidden_layers=2,
                                  num_attention_heads=2).save_pretrained(str(ckpt / "image_encoder"))
    os.makedirs(ckpt / "scheduler")
    json.dump({"_class_name": "EulerDiscreteScheduler", "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
               "num_train_timesteps": 1000, "prediction_type": "v_prediction", "use_karras_sigmas": True, "sigma_min": 0.002,
               "sigma_max": 700.0, "timestep_spacing": "leading", "timestep_type": "continuous", "steps_offset": 1,
               "interpolation_type": "linear", "skip_prk_steps": True}, open(ckpt / "scheduler" / "scheduler_config.json", "w"))
    again = UNetSpatioTemporalConditionModel.from_pretrained(str(ckpt), subfolder="unet")
    assert again.config.in_channels == 9 and all(torch.equal(a, b) for a, b in zip(unet.state_dict().values(), again.state_dict().values()))
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 255, (150, 200, 3), dtype=np.uint8)).save(tmp_path / "img.jpg")
    m = np.zeros((150, 200), dtype=np.uint8)
    m[40:110, 50:150] = 255
    Image.fromarray(m).save(tmp_path / "img_label.jpg")
    cfg = {"pretrained_model_path": str(ckpt), "seed": 3, "output_dir": str(tmp_path / "out"), "iters": 2,
           "validation_data": {"prompt_image": str(tmp_path / "img.jpg"), "prompt": "",
                               "width": 192, "height": 128, "num_frames": 4, "num_inference_steps": 25, "decode_chunk_size": 2,
                               "fps": 7, "motion_bucket_id": 127}}
    yaml.safe_dump(cfg, open(tmp_path / "svd.yaml", "w"))
    results = eval_svd.main(["--config", str(tmp_path / "svd.yaml"), "--eval", "validation_data.num_inference_steps=3"])
    assert len(results) == 2
    _, frames = results[0]
    # 200x150 at a 192x128 pixel budget keeps its aspect ratio in multiples of 64 (train_svd.py:741-745): 192x128
    assert frames.shape == (4, 128, 192, 3) and frames.dtype == np.uint8
    assert not np.array_equal(results[0][1], results[1][1])                                  # per-sample seeds differ
    out = tmp_path / "out" / "img"
    assert (out / "0.gif").exists() and (out / "1.gif").exists() and (out / "0_mask.jpg").exists()
    assert Image.open(out / "0.gif").n_frames == 4
