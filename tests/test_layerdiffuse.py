"""Layerdiffuse RGBA add-ons (animate_anything_amd/layerdiffuse.py) against the oracle restatement of
/root/reference/models/layerdiffuse_VAE.py + the stage-2 encode / alpha-decode call sites, on identical seeded weights:
emulator backend on CPU (tiny frames), MI355X backend under `-m gpu` (384x384 frames, BASELINE.json configs[4])."""
import pytest
import torch

from oracle import layerdiffuse as O
from animate_anything_amd import layerdiffuse as P
from util import rel_err, seeded_state


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def dev(request):
    if request.param == "emu":
        request.getfixturevalue("emu")
        return "cpu"
    assert torch.cuda.is_available()
    return "cuda"


def _pair(cls_o, cls_p, dev):
    torch.manual_seed(0)
    ref = cls_o().eval()
    state = seeded_state(ref, rezero_std=0.05)            # the zero-initialised convs are re-drawn: no vacuous path
    ref.load_state_dict(state)
    net = cls_p().eval()
    assert set(net.state_dict().keys()) == set(state.keys())
    net.load_state_dict(state)
    return ref, net.half().to(dev)


def test_offset_encoder(dev):
    ref, net = _pair(O.LatentTransparencyOffsetEncoder, P.LatentTransparencyOffsetEncoder, dev)
    hw = 384 if dev == "cuda" else 16
    x = torch.rand(2, 4, hw, hw + 8, generator=torch.Generator().manual_seed(1)) * 2 - 1
    with torch.no_grad():
        want = ref(x)
        got = net(x.half().to(dev))
    assert got.shape == want.shape == (2, 4, hw // 8, hw // 8 + 1)
    assert rel_err(got, want) < 2e-2


def test_unet384_alpha_decoder(dev):
    ref, net = _pair(O.UNet384, P.UNet384, dev)
    n, hw = (4, 384) if dev == "cuda" else (2, 16)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(n, 3, hw, hw, generator=g) * 2 - 1
    lat = torch.randn(n, 4, hw // 8, hw // 8, generator=g)
    with torch.no_grad():
        want = ref(x, lat)
        got = net(x.half().to(dev), lat.half().to(dev))
    assert got.shape == want.shape == (n, 4, hw, hw)
    assert rel_err(got, want) < 3e-2
    with torch.no_grad():
        no_lat = ref(x, torch.zeros_like(lat))
    assert rel_err(no_lat, want) > 1e-2                    # the latent injection path is live


def test_decode_rgba_matches_oracle(dev):
    """pipeline_stage2.py:290-318 post-processing: premultiplied foreground, hard-thresholded alpha, uint8 frames."""
    ref, net = _pair(O.UNet384, P.UNet384, dev)
    f, hw = (2, 64) if dev == "cuda" else (2, 16)
    g = torch.Generator().manual_seed(3)
    video = torch.rand(1, 3, f, hw, hw, generator=g) * 2 - 1
    lat = torch.randn(1, 4, f, hw // 8, hw // 8, generator=g)
    with torch.no_grad():
        want = O.decode_rgba(video, lat, ref)
        got = P.decode_rgba(video.half().to(dev), lat.half().to(dev), net)
    for a, b in zip(got, want):
        assert a.shape == b.shape and a.dtype == b.dtype
    assert got[0].shape == (f, hw, hw, 4) and set(torch.from_numpy(got[1]).unique().tolist()) <= {0, 255}
    assert (abs(got[2].astype(int) - want[2].astype(int)) > 3).mean() < 0.01
    assert (got[1] != want[1]).mean() < 0.01               # alpha flips only where the decoder output sits on the threshold
