"""End-to-end host logic on CPU: the product UNet3DConditionModel (token path, packing, grids,
skip bookkeeping, upsample-size forwarding) driven through the SIMT emulator against the oracle."""
import pytest
import torch

import oracle
from animate_anything_amd.unet3d import UNet3DConditionModel
from util import TINY_UNET, rel_err, seeded_state, unet_inputs


@pytest.mark.parametrize("h,w", [(5, 7)])
def test_tiny_unet_matches_oracle(emu, h, w):
    torch.manual_seed(0)
    ref = oracle.UNet3DConditionModel(**TINY_UNET).eval()
    state = seeded_state(ref)
    ref.load_state_dict(state)
    net = UNet3DConditionModel(**TINY_UNET).eval()
    assert set(net.state_dict().keys()) == set(state.keys())
    net.load_state_dict(state)
    net = net.half()
    i = unet_inputs(h=h, w=w, text_len=9)
    with torch.no_grad():
        want = ref(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
        got = net(i["sample"].half(), i["t"], i["text"].half(), i["cond"].half(), i["mask"].half(),
                  motion=i["motion"]).sample
    assert got.shape == want.shape == (2, 4, 2, h, w)
    assert rel_err(got, want) < 3e-2
