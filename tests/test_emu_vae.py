"""Product AutoencoderKL (token path incl. the materialised single-head attention) on the SIMT
emulator against the oracle, CPU only."""
import torch

import oracle
from animate_anything_amd.vae import AutoencoderKL
from util import TINY_VAE, rel_err, seeded_state


def test_tiny_vae_encode_decode(emu):
    torch.manual_seed(0)
    ref = oracle.AutoencoderKL(**TINY_VAE).eval()
    state = seeded_state(ref)
    ref.load_state_dict(state)
    vae = AutoencoderKL(**TINY_VAE).eval()
    assert set(vae.state_dict().keys()) == set(state.keys())
    vae.load_state_dict(state)
    vae = vae.half()
    g = torch.Generator().manual_seed(7)
    x = torch.rand(2, 3, 12, 10, generator=g) * 2 - 1
    with torch.no_grad():
        want_z = ref.encode(x).latent_dist.mode()
        got_z = vae.encode(x.half()).latent_dist.mode()
        assert got_z.shape == want_z.shape == (2, 4, 6, 5)
        assert rel_err(got_z, want_z) < 2e-2
        want_img = ref.decode(want_z).sample
        got_img = vae.decode(want_z.half()).sample
        assert got_img.shape == want_img.shape == (2, 3, 12, 10)
        assert rel_err(got_img, want_img) < 2e-2
