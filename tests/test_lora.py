"""LoRA fold (animate_anything_amd/lora.py) against the reference's injected forward (oracle/lora.py restatement of
/root/reference/utils/lora.py): same adapters, folded weights == wrapped layers, for both file layouts (adapters on every
Linear/Conv layer; adapters only on diffusers-0.24's plain torch.nn layers)."""
import pytest
import torch

import oracle
from oracle import lora as olora
from animate_anything_amd import lora as L
from animate_anything_amd.unet3d import UNet3DConditionModel
from util import TINY_UNET, rel_err, seeded_state, unet_inputs


def make_loras(targets, r=4, seed=5):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _, m in targets:
        w = m.weight
        out.append(torch.randn((w.shape[0], r) + (1,) * (w.dim() - 2), generator=g) * 0.2)          # lora_up   [out, r, 1..]
        out.append(torch.randn((r,) + tuple(w.shape[1:]), generator=g) * 0.2 / w[0].numel() ** 0.5)  # lora_down [r, in, k..]
    return out


def _models():
    torch.manual_seed(0)
    ref = oracle.UNet3DConditionModel(**TINY_UNET).eval()
    state = seeded_state(ref)
    ref.load_state_dict(state)
    net = UNet3DConditionModel(**TINY_UNET).eval()
    net.load_state_dict(state)
    return ref, net


@pytest.mark.parametrize("mode", ["plain", "all"])
def test_fold_equals_injected_forward_fp32(mode):
    """Weight-space identity on CPU in fp32 (no kernels involved): fold into one oracle copy, inject into another."""
    ref, _ = _models()
    folded, _ = _models()
    folded.__class__.__name__ = "UNet3DConditionModel"
    every = L._candidates(folded, L.UNET_REPLACE)
    targets = every if mode == "all" else [(n, m) for n, m in every if L._PLAIN_024.match(n)]
    assert 0 < len(targets) <= len(every)
    assert any("temp_convs" in n for n, _ in targets) and any(n == "conv_in2" for n, _ in targets)
    loras = make_loras(targets)
    names = L.fold_lora_(folded, loras, scale=0.7)
    assert names == [n for n, _ in targets]
    olora.inject(ref, names, loras, scale=0.7)
    i = unet_inputs(h=5, w=6, text_len=9)
    with torch.no_grad():
        want = ref(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
        got = folded(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
    assert rel_err(got, want) < 1e-4
    base, _ = _models()
    with torch.no_grad():
        plain = base(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
    assert rel_err(plain, want) > 1e-2                      # the adapters really change the output


def test_product_unet_with_folded_lora_matches_injected_oracle(emu, tmp_path):
    """The reference entry point (`inject_inferable_lora(pipeline, lora_path)`) on the product UNet through the HIP token
    path (emulator): forward once (builds the packed / fused weight caches), fold, forward again."""
    ref, net = _models()
    net = net.half()
    i = unet_inputs(h=5, w=6, text_len=9)
    call = lambda: net(i["sample"].half(), i["t"], i["text"].half(), i["cond"].half(), i["mask"].half(), motion=i["motion"]).sample
    with torch.no_grad():
        before = call()
    targets = L.lora_targets(net, len([1 for n, _ in L._candidates(net, L.UNET_REPLACE) if L._PLAIN_024.match(n)]))
    loras = make_loras(targets)
    torch.save(loras, tmp_path / "100_unet.pt")
    torch.save([torch.zeros(1)], tmp_path / "notes.pt")

    class Pipe:
        unet, text_encoder = net, None
    done = L.inject_inferable_lora(Pipe, str(tmp_path), r=4)
    assert done["unet"] == [n for n, _ in targets]
    olora.inject(ref, done["unet"], loras)
    with torch.no_grad():
        want = ref(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
        got = call()
    assert rel_err(got, want) < 3e-2
    assert rel_err(before, want) > 5e-2                     # stale caches would reproduce `before`


def test_unknown_layout_and_bad_shapes_are_rejected():
    _, net = _models()
    with pytest.raises(ValueError):
        L.fold_lora_(net, [torch.zeros(3, 2), torch.zeros(2, 3)] * 3)
    targets = L.lora_targets(net, len(L._candidates(net, L.UNET_REPLACE)))
    loras = make_loras(targets)
    loras[1] = loras[1][:, :-1]                             # wrong input width for the first layer
    with pytest.raises(ValueError):
        L.fold_lora_(net, loras)


def test_text_encoder_fold_matches_injected_clip():
    transformers = pytest.importorskip("transformers")
    cfg = transformers.CLIPTextConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                      vocab_size=100, max_position_embeddings=16)
    torch.manual_seed(0)
    a = transformers.CLIPTextModel(cfg).eval()
    b = transformers.CLIPTextModel(cfg).eval()
    b.load_state_dict(a.state_dict())
    targets = L._candidates(a, L.TEXT_ENCODER_REPLACE)
    assert len(targets) == 2 * 6                            # q, k, v, out_proj, fc1, fc2 per layer
    loras = make_loras(targets, r=3)
    L.fold_lora_(a, loras, L.TEXT_ENCODER_REPLACE, mode="all")
    olora.inject(b, [n for n, _ in targets], loras)
    ids = torch.randint(0, 100, (2, 16))
    with torch.no_grad():
        assert rel_err(a(ids)[0], b(ids)[0]) < 1e-4
