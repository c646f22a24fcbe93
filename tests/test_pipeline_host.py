"""Host logic on CPU: the product scheduler against the oracle's independent restatement, and the
product denoising loop (CFG batching, fused guidance + DPM-Solver++ kernel via the emulator, timestep
handling, return convention) against the oracle pipeline, both driving the same stand-in UNet."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import oracle
from animate_anything_amd.pipeline import LatentToVideoPipeline, calculate_latent_motion_score, tensor2vid
from animate_anything_amd.schedulers import DDPMScheduler, DPMSolverMultistepScheduler


@pytest.mark.parametrize("steps,spacing", [(25, "leading"), (10, "leading"), (25, "linspace")])
def test_dpm_solver_matches_oracle(steps, spacing):
    a = DPMSolverMultistepScheduler(timestep_spacing=spacing, steps_offset=1 if spacing == "leading" else 0)
    b = oracle.DPMSolverMultistepScheduler(timestep_spacing=spacing, steps_offset=1 if spacing == "leading" else 0)
    a.set_timesteps(steps)
    b.set_timesteps(steps)
    assert a.timesteps.tolist() == b.timesteps.tolist()
    assert np.allclose(a.sigmas, b.sigmas)
    g = torch.Generator().manual_seed(0)
    xa = xb = torch.randn(3, 4, 8, 8, generator=g)
    for t in a.timesteps:
        eps = torch.randn(3, 4, 8, 8, generator=g)
        xa = a.step(eps, t, xa).prev_sample
        xb = b.step(eps, t, xb)
        assert (xa - xb).abs().max() < 1e-4
    if spacing == "leading":
        assert a.timesteps[0] == (1000 // (steps + 1)) * steps + 1


def test_add_noise_forms_agree():
    s = DPMSolverMultistepScheduler()
    s.set_timesteps(25)
    g = torch.Generator().manual_seed(1)
    x0, n = torch.randn(2, 4, 3, 5, 5, generator=g), torch.randn(2, 4, 3, 5, 5, generator=g)
    t = torch.tensor([int(s.timesteps[0])] * 2)
    via_sigma = s.add_noise(x0, n, t)
    via_ddpm = DDPMScheduler().add_noise(x0, n, t)
    ref = oracle.ddpm_add_noise(x0, n, int(t[0]))
    assert (via_sigma - ref).abs().max() < 1e-5 and (via_ddpm - ref).abs().max() < 1e-5


class _StubUNet:
    def __init__(self, dtype):
        self.dtype = dtype

    def __call__(self, x, t, encoder_hidden_states=None, condition_latent=None, mask=None, motion=None, **_):
        txt = encoder_hidden_states.float().mean(dim=(1, 2)).reshape(-1, 1, 1, 1, 1)
        y = 0.3 * x.float() * float(np.cos(int(t) / 300.0)) + 0.1 * condition_latent.float() + txt \
            + 0.05 * mask.float() * float(motion.float().sum())
        return SimpleNamespace(sample=y.to(x.dtype))


@pytest.mark.parametrize("guidance", [9.0, 1.0])
def test_denoise_loop_matches_oracle(emu, guidance):
    g = torch.Generator().manual_seed(2)
    r = lambda *s: torch.randn(*s, generator=g)
    lat, cond, mask = r(1, 4, 3, 4, 4), r(1, 4, 1, 4, 4), (r(1, 1, 1, 4, 4) > 0).float()
    pos, neg = r(1, 7, 16), r(1, 7, 16)
    want = oracle.LatentToVideoPipeline(None, _StubUNet(torch.float32), oracle.DPMSolverMultistepScheduler())(
        latents=lat, prompt_embeds=pos, negative_prompt_embeds=neg, condition_latent=cond, mask=mask, motion=[3.0],
        num_inference_steps=6, guidance_scale=guidance, return_dict=False)[1]
    pipe = LatentToVideoPipeline(vae=None, unet=_StubUNet(torch.float16), scheduler=DPMSolverMultistepScheduler())
    frames, got = pipe(latents=lat, prompt_embeds=pos.half(), negative_prompt_embeds=neg.half(),
                       condition_latent=cond.half(), mask=mask.half(), motion=[3.0], num_inference_steps=6,
                       guidance_scale=guidance, return_dict=False)
    assert frames is None and got.shape == lat.shape
    assert (got.float() - want).abs().max() < 2e-2 * want.abs().max()


def test_pipeline_input_checks_and_helpers():
    pipe = LatentToVideoPipeline(vae=None, unet=_StubUNet(torch.float16))
    with pytest.raises(ValueError):
        pipe(latents=torch.zeros(1, 4, 2, 4, 4), height=30, width=32, prompt_embeds=torch.zeros(1, 7, 16))
    with pytest.raises(ValueError):
        pipe(latents=torch.zeros(1, 4, 2, 4, 4))
    v = torch.linspace(-1, 1, 2 * 3 * 2 * 4 * 5).reshape(2, 3, 2, 4, 5)
    fr, ofr = tensor2vid(v), oracle.tensor2vid(v)
    assert len(fr) == 2 and fr[0].shape == (4, 10, 3) and all((a == b).all() for a, b in zip(fr, ofr))
    z = torch.randn(2, 4, 5, 3, 3)
    assert torch.allclose(calculate_latent_motion_score(z), 10 * (z[:, :, 1:] - z[:, :, :-1]).abs().mean(dim=[2, 3, 4]).sum(1))


def test_pipeline_save_pretrained_roundtrip(tmp_path):
    """LatentToVideoPipeline.save_pretrained writes the diffusers directory layout from_pretrained reads (reference
    train.py:298-299, 799-804): identical weights, configs and scheduler settings after a round trip (no kernels involved)."""
    import torch
    from animate_anything_amd.pipeline import LatentToVideoPipeline
    from animate_anything_amd.schedulers import DPMSolverMultistepScheduler
    from animate_anything_amd.unet3d import UNet3DConditionModel
    from animate_anything_amd.vae import AutoencoderKL
    from util import TINY_UNET, TINY_VAE
    torch.manual_seed(0)
    pipe = LatentToVideoPipeline(vae=AutoencoderKL(**TINY_VAE), unet=UNet3DConditionModel(**TINY_UNET),
                                 scheduler=DPMSolverMultistepScheduler(beta_end=0.013))
    pipe.save_pretrained(str(tmp_path / "ckpt"))
    again = LatentToVideoPipeline.from_pretrained(str(tmp_path / "ckpt"))
    assert again.unet.config.block_out_channels == tuple(TINY_UNET["block_out_channels"]) and again.unet.motion_mask
    assert again.scheduler.config.beta_end == 0.013 and again.vae_scale_factor == pipe.vae_scale_factor
    for a, b in ((pipe.unet, again.unet), (pipe.vae, again.vae)):
        assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))


def test_checkpoint_lookup_rejects_path_variants_and_unpickles_tensors_only(tmp_path):
    """a caller-supplied `variant` is spliced into the file name - tags only; .bin checkpoints load with
    weights_only=True (a pickled object that is not a tensor container is refused)."""
    import pickle
    import pytest
    import torch
    from animate_anything_amd._ckpt import load_state
    torch.save({"w": torch.ones(2, 3)}, tmp_path / "diffusion_pytorch_model.fp16.bin")
    sd = load_state(str(tmp_path), "diffusion_pytorch_model", variant="fp16")
    assert sd["w"].shape == (2, 3)
    for bad in ("../x", "a/b", "fp16.bin\x00"):
        with pytest.raises(ValueError):
            load_state(str(tmp_path), "diffusion_pytorch_model", variant=bad)

    class Evil:
        def __reduce__(self):
            return (print, ("code ran",))
    (tmp_path / "evil").mkdir()
    with open(tmp_path / "evil" / "m.bin", "wb") as f:
        pickle.dump({"w": Evil()}, f)
    with pytest.raises(Exception):
        load_state(str(tmp_path / "evil"), "m")
    with pytest.raises(FileNotFoundError):
        load_state(str(tmp_path), "nothing_here")


def test_bench_quotes_pmc_traffic_only_for_the_library_it_was_measured_on(tmp_path):
    """bench.py `roofline.traffic` (VERDICT r04 weak point 12): the committed PMC record counts only for the library SOURCES it names and the
    same number of contraction launches per step; a record of another library, or none, gives null with the reason."""
    import json
    import os
    import bench
    from animate_anything_amd import build
    sid = build.source_id()
    assert len(sid) == 16 and sid == build.source_id()
    rec = {"contraction_kernels": {"hbm_bytes_per_launch": 123456, "launches_per_step": 443}, "library_source_sha256_16": sid}
    (tmp_path / "r05_traffic_pmc.json").write_text(json.dumps(rec))
    assert bench.pick_traffic_record(str(tmp_path), sid, 443) == (123456, "r05_traffic_pmc.json", 443, None)
    got = bench.pick_traffic_record(str(tmp_path), sid, 444)
    assert got[0] is None and "443 contraction launches" in got[3]
    got = bench.pick_traffic_record(str(tmp_path), "0" * 16, 443)
    assert got[0] is None and "measured on library sources" in got[3]
    (tmp_path / "r06_traffic_pmc.json").write_text(json.dumps({"contraction_kernels": {"hbm_bytes_per_launch": 1, "launches_per_step": 443}}))
    got = bench.pick_traffic_record(str(tmp_path), sid, 443)           # the NEWEST record decides: an unkeyed one is not trusted
    assert got[0] is None and got[1] == "r06_traffic_pmc.json"
    assert bench.pick_traffic_record(str(tmp_path / "none"), sid, 443) == (None, None, None, None)


def test_committed_pmc_record_names_the_committed_sources():
    """The newest PMC record under profiles/ was measured on the library sources of this tree (a closing-run artefact: scripts/pmc_traffic.sh
    rewrites it).  While the sources are ahead of the record - kernel work between two closing runs - bench.py prints `traffic: null` with the
    reason; that state is reported here as a SKIP naming the two source ids, not hidden and not a failure of the library."""
    import json
    import os
    import bench
    import pytest
    from animate_anything_amd import build
    sid = build.source_id()
    prof = os.path.join(bench.ROOT, "profiles")
    names = sorted(n for n in os.listdir(prof) if n.endswith("traffic_pmc.json"))
    assert names, "profiles/: no PMC traffic record at all"
    rec = json.load(open(os.path.join(prof, names[-1])))
    launches = rec["contraction_kernels"]["launches_per_step"]
    got = bench.pick_traffic_record(prof, sid, launches)
    if not got[0]:
        pytest.skip(f"profiles/{names[-1]} was measured on library sources {rec.get('library_source_sha256_16')}, this tree is {sid}: {got[3]}")
    assert got[0] > 0 and got[2] == launches
