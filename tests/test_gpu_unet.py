"""GPU parity of the product UNet3DConditionModel (libaa_mi355.so on the MI355X) against the CPU oracle
on identical seeded weights and inputs (fp16 storage, fp32 accumulate; tolerance: latent MSE < 1e-3
and max-normalised error < 3e-2, the north-star fp16 bar)."""
import pytest
import torch

import oracle
from animate_anything_amd.unet3d import UNet3DConditionModel
from util import SMALL_UNET, rel_err, seeded_state, unet_inputs

pytestmark = pytest.mark.gpu


def _pair(cfg):
    torch.manual_seed(0)
    ref = oracle.UNet3DConditionModel(**cfg).eval()
    state = seeded_state(ref)
    ref.load_state_dict(state)
    net = UNet3DConditionModel(**cfg).eval()
    net.load_state_dict(state)
    return ref, net


def _run(ref, net, i, dtype, graph=False):
    net = net.to(dtype).cuda()
    if graph:
        net.enable_graph()
    dev = lambda x: x.to(dtype).cuda()
    with torch.no_grad():
        want = ref(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
        for _ in range(2 if graph else 1):
            got = net(dev(i["sample"]), i["t"], dev(i["text"]), dev(i["cond"]), dev(i["mask"]), motion=i["motion"]).sample
    torch.cuda.synchronize()
    return got.float().cpu(), want


@pytest.mark.parametrize("h,w,frames", [(16, 16, 3), (11, 14, 2)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_small_unet(h, w, frames, dtype):
    ref, net = _pair(SMALL_UNET)
    got, want = _run(ref, net, unet_inputs(b=2, frames=frames, h=h, w=w, text_dim=128), dtype)
    tol = 3e-2 if dtype == torch.float16 else 1.5e-1
    assert rel_err(got, want) < tol
    assert ((got - want) ** 2).mean().item() < (1e-3 if dtype == torch.float16 else 1e-2)


def test_small_unet_graph_replay_matches_eager():
    ref, net = _pair(SMALL_UNET)
    i = unet_inputs(b=2, frames=3, h=16, w=16, text_dim=128)
    eager, want = _run(ref, net, i, torch.float16)
    replay, _ = _run(ref, net, i, torch.float16, graph=True)
    assert torch.equal(eager, replay)


def test_full_architecture_unet():
    """The v1.02 architecture (320/640/1280/1280, 1413M parameters) at a small spatial size."""
    cfg = dict(motion_mask=True, motion_strength=True)
    ref, net = _pair(cfg)
    got, want = _run(ref, net, unet_inputs(b=2, frames=2, h=8, w=8, text_dim=1024), torch.float16)
    assert rel_err(got, want) < 3e-2
    assert ((got - want) ** 2).mean().item() < 1e-3
