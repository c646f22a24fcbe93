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


# ---------------------------------------------------------------------------------------------------------------------------
# The reference's own LoRA code as the checker: /root/reference/utils/lora.py imports cleanly here.
# `inject_trainable_lora_extended` (:433-479) builds the wrappers, `save_lora_weight` (:569-581) writes the file,
# `monkeypatch_or_replace_lora_extended` (:861-977) is what `inject_inferable_lora` (:482-526) runs at inference.
import refload  # noqa: E402

needs_ref = pytest.mark.skipif(not refload.have_reference(), reason="reference checkout not present")


def _reference_lora_file(tmp_path, mark_compatible=False, r=4, seed=11):
    """A LoRA file written by the REFERENCE for the oracle UNet, and the reference-injected model that produced it.
    mark_compatible: give the layers that diffusers 0.24 builds from LoRACompatibleLinear / LoRACompatibleConv a subclass
    type, so the reference skips them exactly as it does on a real diffusers model (utils/lora.py:963-965)."""
    R = refload.load("utils/lora.py")
    model, _ = _models()
    if mark_compatible:
        sub = {torch.nn.Linear: type("LoRACompatibleLinear", (torch.nn.Linear,), {}),
               torch.nn.Conv2d: type("LoRACompatibleConv", (torch.nn.Conv2d,), {})}
        for n, m in model.named_modules():
            if type(m) in sub and not L._PLAIN_024.match(n):
                m.__class__ = sub[type(m)]
    R.inject_trainable_lora_extended(model, target_replace_module={"UNet3DConditionModel"}, r=r)
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (R.LoraInjectedLinear, R.LoraInjectedConv2d, R.LoraInjectedConv3d)):
            m.lora_up.weight.data = torch.randn(m.lora_up.weight.shape, generator=g) * 0.2      # (zero-initialised by the reference)
            m.dropout = torch.nn.Identity()
    path = str(tmp_path / "unet.pt")
    R.save_lora_weight(model, path, target_replace_module={"UNet3DConditionModel"})
    return R, model.eval(), path


@needs_ref
@pytest.mark.parametrize("mark_compatible", [False, True])
def test_fold_equals_the_references_own_injection(tmp_path, mark_compatible):
    """File written by the reference -> (a) the reference's inference-time loader on a fresh oracle UNet, (b) the restated
    wrapper of oracle/lora.py, (c) the product's fold: all three must be the function the reference trained."""
    R, trained, path = _reference_lora_file(tmp_path, mark_compatible)
    i = unet_inputs(h=5, w=6, text_len=9)
    run = lambda m: m(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
    with torch.no_grad():
        want = run(trained)
    # (a) reference loader
    a, _ = _models()
    if mark_compatible:
        for (n, m), (_, t) in zip(a.named_modules(), _models()[0].named_modules()):
            pass
        sub = {torch.nn.Linear: type("LoRACompatibleLinear", (torch.nn.Linear,), {}),
               torch.nn.Conv2d: type("LoRACompatibleConv", (torch.nn.Conv2d,), {})}
        for n, m in a.named_modules():
            if type(m) in sub and not L._PLAIN_024.match(n):
                m.__class__ = sub[type(m)]
    R.monkeypatch_or_replace_lora_extended(a, torch.load(path), target_replace_module={"UNet3DConditionModel"}, r=4)
    for m in a.modules():
        if hasattr(m, "dropout") and hasattr(m, "lora_up"):
            m.dropout = torch.nn.Identity()
    with torch.no_grad():
        assert rel_err(run(a.eval()), want) < 1e-5
    # (c) product fold (weight space) on a plain oracle copy
    c, _ = _models()
    names = L.fold_lora_(c, torch.load(path))
    with torch.no_grad():
        assert rel_err(run(c), want) < 1e-4
    # (b) the restated wrapper on the layers the fold reported
    b, _ = _models()
    olora.inject(b, names, torch.load(path))
    with torch.no_grad():
        assert rel_err(run(b), want) < 1e-5
    base, _ = _models()
    with torch.no_grad():
        assert rel_err(run(base), want) > 1e-2


@needs_ref
def test_traversal_order_is_the_references(tmp_path):
    """The adapter order in a reference-written file is the order `fold_lora_` walks: down_blocks, up_blocks, mid_block
    (the registration order of the reference's UNet is part of the file format)."""
    R, trained, path = _reference_lora_file(tmp_path)
    net = UNet3DConditionModel(**TINY_UNET).eval()
    cand = [n for n, _ in L._candidates(net, L.UNET_REPLACE)]
    ref_names = []
    for parent, name, child in R._find_modules(trained, {"UNet3DConditionModel"}, search_class=[R.LoraInjectedLinear, R.LoraInjectedConv2d, R.LoraInjectedConv3d]):
        full = [n for n, m in trained.named_modules() if m is child]
        ref_names.append(full[0])
    assert cand == ref_names
    first = lambda pre: next(k for k, n in enumerate(cand) if n.startswith(pre))
    assert first("down_blocks") < first("up_blocks") < first("mid_block") < first("conv_out")


@needs_ref
def test_collapse_lora_is_the_fold(tmp_path):
    """`collapse_lora` (utils/lora.py:780-815) folds W += alpha * up @ down in place for the wrappers below its target classes
    (Attention / ResnetBlock2D / GEGLU ...): wherever it acted, the result is the product's folded weight."""
    R, trained, path = _reference_lora_file(tmp_path)
    folded, _ = _models()
    L.fold_lora_(folded, torch.load(path), scale=1.0)
    base = dict(_models()[0].named_parameters())
    R.collapse_lora(trained, alpha=1.0)
    got = dict(folded.named_parameters())
    collapsed = 0
    for n, m in trained.named_modules():
        if isinstance(m, (R.LoraInjectedLinear, R.LoraInjectedConv2d, R.LoraInjectedConv3d)):
            w = m.linear.weight if isinstance(m, R.LoraInjectedLinear) else m.conv.weight
            if torch.equal(w, base[n + ".weight"]):
                continue                                    # not below one of collapse_lora's target classes
            assert torch.allclose(w, got[n + ".weight"], atol=1e-5), n
            collapsed += 1
    assert collapsed > 10


@pytest.mark.gpu
def test_gpu_fold_matches_the_reference_written_golden():
    """`-m gpu`: the product UNet on the GPU (fp16, graph on), one forward to build its packed-weight /
    graph caches, then `inject_inferable_lora` on a file whose adapters and expected output were written by the REFERENCE's
    own utils/lora.py on the oracle (tests/golden/make_lora_golden.py) - the stale-cache path included."""
    import os
    from util import SMALL_UNET
    gold = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lora_small_unet.pt"))
    torch.manual_seed(0)
    ref = oracle.UNet3DConditionModel(**SMALL_UNET).eval()
    state = seeded_state(ref)
    net = UNet3DConditionModel(**SMALL_UNET).eval()
    net.load_state_dict(state)
    net = net.half().cuda()
    net.enable_graph()
    i = unet_inputs(**gold["inputs"])
    dev = lambda t: t.half().cuda()
    call = lambda: net(dev(i["sample"]), i["t"], dev(i["text"]), dev(i["cond"]), dev(i["mask"]), motion=i["motion"]).sample.float().cpu()
    with torch.no_grad():
        before = call()
        before2 = call()                                    # (graph replay)
    assert rel_err(before2, before) < 1e-3

    class Pipe:
        unet, text_encoder = net, None
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        torch.save([t.float() for t in gold["loras"]], os.path.join(d, "1000_unet.pt"))
        done = L.inject_inferable_lora(Pipe, d, r=gold["r"])
    assert len(done["unet"]) == len(gold["loras"]) // 2
    with torch.no_grad():
        got = call()
        got2 = call()
    want = gold["expected"]
    assert rel_err(got, want) < 3e-2 and rel_err(got2, want) < 3e-2
    assert rel_err(before, want) > 5e-2                     # stale caches would reproduce `before`
