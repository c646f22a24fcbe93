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


def test_weight_caches_follow_load_state_dict_and_in_place_updates(emu):
    """the fused copies of the weights (Q|K|V, GEGLU, the batched time-embedding / text K|V packs) must be
    rebuilt after load_state_dict() and after in-place parameter edits - forward, load new weights, forward, compare
    with the oracle carrying the same new weights."""
    torch.manual_seed(0)
    ref = oracle.UNet3DConditionModel(**TINY_UNET).eval()
    net = UNet3DConditionModel(**TINY_UNET).eval().half()
    i = unet_inputs(h=5, w=6, text_len=9)

    def both(state):
        ref.load_state_dict(state)
        net.load_state_dict(state)                     # copies into the existing fp16 parameters in place
        with torch.no_grad():
            want = ref(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
            got = net(i["sample"].half(), i["t"], i["text"].half(), i["cond"].half(), i["mask"].half(),
                      motion=i["motion"]).sample
        return rel_err(got, want), want

    e1, w1 = both(seeded_state(ref, seed=0))
    e2, w2 = both(seeded_state_other(ref))
    assert e1 < 3e-2 and e2 < 3e-2, (e1, e2)
    assert rel_err(w1, w2) > 0.1                        # the two checkpoints really differ
    # in-place edit of one fused operand family without load_state_dict
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_.endswith(("to_q.weight", "ff.net.0.proj.weight", "time_emb_proj.weight", "attn2.to_k.weight")):
                p_.mul_(0.5)
        ref.load_state_dict({k: v.float() for k, v in net.state_dict().items()})
        want = ref(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
        got = net(i["sample"].half(), i["t"], i["text"].half(), i["cond"].half(), i["mask"].half(), motion=i["motion"]).sample
    assert rel_err(got, want) < 3e-2


def seeded_state_other(module):
    g = torch.Generator().manual_seed(77)
    return {k: (torch.randn(v.shape, generator=g) * (0.05 if v.dim() > 1 else 0.3) + (1.0 if "norm" in k and k.endswith("weight") else 0.0))
            for k, v in module.state_dict().items()}


def test_cfg_shared_prefix_matches_full_batch(emu):
    """Under guidance the text-independent prefix of the UNet (conv_in ... first spatial self-attention) is computed for one
    half of the batch and replicated (UNet3DConditionModel._core cfg_dup): same latents as running both halves in full."""
    from animate_anything_amd.pipeline import LatentToVideoPipeline
    from animate_anything_amd.schedulers import DPMSolverMultistepScheduler
    torch.manual_seed(0)
    net = UNet3DConditionModel(**TINY_UNET).eval()
    net.load_state_dict(seeded_state(oracle.UNet3DConditionModel(**TINY_UNET)))
    net = net.half()
    g = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=g)
    lat, cond, pos, neg = r(1, 4, 2, 5, 6), r(1, 4, 1, 5, 6), r(1, 9, 64), r(1, 9, 64)
    mask = torch.zeros(1, 1, 1, 5, 6)
    mask[..., 1:4, 2:5] = 1
    outs = []
    for shared in (True, False):
        pipe = LatentToVideoPipeline(vae=None, unet=net, scheduler=DPMSolverMultistepScheduler())
        pipe.cfg_shared_prefix = shared
        pipe.scheduler.set_timesteps(2)
        with torch.no_grad():
            outs.append(pipe.denoise(lat, torch.cat([neg, pos]).half(), cond.half(), mask.half(), [3.0],
                                     [int(t) for t in pipe.scheduler.timesteps], 9.0))
    assert torch.isfinite(outs[0]).all()
    assert rel_err(outs[0], outs[1]) < 2e-3


@pytest.mark.parametrize("fold_ff", [False, True])
def test_layernorm_fold_replaces_the_layernorm_kernels(emu, monkeypatch, fold_ff):
    """layers.LN_FOLD: with a tile that emits row statistics forced onto every contraction it fits (the 128 x 128 hand-scheduled
    tile), the transformer blocks of the UNet run WITHOUT LayerNorm launches where the producer emitted statistics, and the
    result equals the LayerNorm-kernel form of the same network (and the oracle)."""
    from animate_anything_amd import layers, ops
    torch.manual_seed(0)
    ref = oracle.UNet3DConditionModel(**TINY_UNET).eval()
    state = seeded_state(ref)
    ref.load_state_dict(state)
    net = UNet3DConditionModel(**TINY_UNET).eval()
    net.load_state_dict(state)
    net = net.half()
    i = unet_inputs(h=6, w=6, text_len=9)
    calls = {"ln": 0, "folded": 0}
    real_ln, real_cg = ops.layernorm, ops.conv_gemm

    def counting_ln(*a, **k):
        calls["ln"] += 1
        return real_ln(*a, **k)

    def counting_cg(*a, **k):
        calls["folded"] += k.get("ln_stats") is not None
        return real_cg(*a, **k)
    monkeypatch.setattr(ops, "layernorm", counting_ln)
    monkeypatch.setattr(ops, "conv_gemm", counting_cg)

    def run():
        calls["ln"] = calls["folded"] = 0
        with torch.no_grad():
            return net(i["sample"].half(), i["t"], i["text"].half(), i["cond"].half(), i["mask"].half(), motion=i["motion"]).sample
    lib = __import__("animate_anything_amd._lib", fromlist=["get"]).get()
    lib.aa_set_tile_override(47)                       # preference: taken wherever the packed width is a multiple of 128
    try:
        monkeypatch.setattr(layers, "LN_FOLD", True)
        monkeypatch.setattr(layers, "LN_FOLD_FF", fold_ff)       # norm3 -> GEGLU folded as well (off by default: measured slower)
        folded = run()
        n_ln_fold, n_folded = calls["ln"], calls["folded"]
        monkeypatch.setattr(layers, "LN_FOLD", False)
        plain = run()
        n_ln_plain = calls["ln"]
    finally:
        lib.aa_set_tile_override(-1)
    assert n_folded > 0 and n_ln_fold + n_folded == n_ln_plain and calls["folded"] == 0
    with torch.no_grad():
        want = ref(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
    assert rel_err(folded, want) < 3e-2 and rel_err(plain, want) < 3e-2 and rel_err(folded, plain) < 2e-2


@pytest.mark.parametrize("seed,splits,tickets", [(1, False, False), (2, False, False), (3, False, False), (4, True, False), (5, True, True)])
def test_random_tile_assignments_tiny_unet(emu, seed, splits, tickets):
    """Tile fuzzing on the emulator (the CPU twin of tests/test_gpu_fullsize.py::test_random_tile_assignments_at_the_metric_configuration,
    ops.TILE_PICKER): every contraction of the tiny UNet runs on a random eligible tile of the library's table - compiled, halo-slab and
    hand-scheduled ones, with and without row statistics / a folded LayerNorm - and the forward still matches the oracle.  (Register
    allocation bugs of the real code object are the GPU test's and the build audit's business; this one covers the index arithmetic
    of every tile x epilogue form the graph can reach.)"""
    import random
    from animate_anything_amd import ops
    torch.manual_seed(0)
    ref = oracle.UNet3DConditionModel(**TINY_UNET).eval()
    state = seeded_state(ref)
    ref.load_state_dict(state)
    net = UNet3DConditionModel(**TINY_UNET).eval()
    net.load_state_dict(state)
    net = net.half()
    i = unet_inputs(h=6, w=6, text_len=9)
    rng = random.Random(seed)
    used = {}

    def pick(key, cands):
        if key not in used:
            tile, sp = rng.choice(cands)
            if splits:                                          # `splits`: random K splits on top (the library clamps them to the K steps there
                sp = rng.choice((0, 0, 2, 3, 5))                #  are and ignores them where a call cannot split): partials + reduce launch, or -
            used[key] = (tile, sp)                              #  `tickets` - the finish inside the kernel on the tiles that carry it
        return used[key]
    ops.TILE_PICKER, keep = pick, ops.USE_TICKETS
    ops.USE_TICKETS = tickets
    try:
        with torch.no_grad():
            got = net(i["sample"].half(), i["t"], i["text"].half(), i["cond"].half(), i["mask"].half(), motion=i["motion"]).sample
    finally:
        ops.TILE_PICKER, ops.USE_TICKETS = None, keep
    with torch.no_grad():
        want = ref(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
    assert len({c[0] for c in used.values()}) >= 8, used          # a spread of tiles really ran
    assert rel_err(got, want) < 3e-2, (rel_err(got, want), sorted(used.items(), key=str))


@pytest.mark.parametrize("temporal", [False, True])
def test_ff_out_and_proj_out_as_one_contraction(emu, monkeypatch, temporal):
    """layers.FF_PROJ_MERGE: proj_out(ff_out(h) + x) + residual of Transformer2DModel / TransformerTemporalModel
    (oracle/layers.py:205-284; diffusers BasicTransformerBlock's FeedForward followed by the wrapper's proj_out) as ONE
    two-source contraction over [h | x] with the weights [Wp W2 | Wp]: against the oracle module, against the two-call form, two
    contraction launches fewer per transformer, and rebuilt when either weight changes in place."""
    from animate_anything_amd import layers as L, ops
    torch.manual_seed(3)
    C, heads, g = 128, 2, L.Grid(2, 3, 4, 5)
    if temporal:
        ref, net = oracle.TransformerTemporalModel(heads, 64, C), L.TransformerTemporalModel(heads, 64, C)
    else:
        ref, net = oracle.Transformer2DModel(heads, 64, C, cross_attention_dim=64), L.Transformer2DModel(heads, 64, C, cross_attention_dim=64)
    ref = ref.eval()
    state = seeded_state(ref)
    ref.load_state_dict(state)
    net.load_state_dict(state)
    net = net.half().eval()
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(g.tokens, C, generator=gen)
    text = torch.randn(g.clips * 7, 64, generator=gen)

    def run():
        ops.TRACE = []
        with torch.no_grad():
            y = net.tokens(x.half(), g) if temporal else net.tokens(x.half(), g, text.half(), 7)
        n, ops.TRACE = len(ops.TRACE), None
        return y.float(), n

    def want():
        x5 = x.reshape(g.clips, g.frames, g.h, g.w, C)
        with torch.no_grad():
            if temporal:
                y = ref(x5.permute(0, 1, 4, 2, 3).reshape(g.images, C, g.h, g.w), num_frames=g.frames).sample
            else:
                y = ref(x5.permute(0, 1, 4, 2, 3).reshape(g.images, C, g.h, g.w),
                        encoder_hidden_states=text.reshape(g.clips, 7, 64).repeat_interleave(g.frames, 0)).sample
        return y.permute(0, 2, 3, 1).reshape(g.tokens, C)

    merged, n_merged = run()
    monkeypatch.setattr(L, "FF_PROJ_MERGE", False)
    two_calls, n_two = run()
    monkeypatch.setattr(L, "FF_PROJ_MERGE", True)
    w = want()
    assert n_two - n_merged == 1                               # ff-out and proj_out are one launch
    assert rel_err(merged, w) < 2e-2 and rel_err(two_calls, w) < 2e-2
    assert rel_err(merged, two_calls) < 1e-2
    with torch.no_grad():                                      # an in-place edit of either weight rebuilds the merged pack
        net.proj_out.weight.mul_(0.5)
        net.transformer_blocks[0].ff.net[2].bias.add_(0.25)
        ref.load_state_dict({k: v.float() for k, v in net.state_dict().items()})
    again, _ = run()
    assert rel_err(again, want()) < 2e-2 and rel_err(again, merged) > 0.05


@pytest.mark.parametrize("temporal", [False, True])
def test_k_equals_c_projections_with_the_rows_in_registers(emu, monkeypatch, temporal):
    """layers.LINEAR_ROWS: proj_in, norm1 -> Q|K|V, to_out + residual, norm2 -> to_q, to_out + residual of a 320-channel transformer on
    ops.linear_rows (oracle/layers.py:205-284): against the oracle module and against the contraction form (LayerNorm folded from
    producer-written statistics); five (spatial) / three (temporal: Q|K|V live in ops.seq_self_attention) contraction launches fewer."""
    from animate_anything_amd import layers as L, ops
    torch.manual_seed(4)
    C, heads, g = 320, 5, L.Grid(1, 3, 3, 5)
    if temporal:
        ref, net = oracle.TransformerTemporalModel(heads, 64, C), L.TransformerTemporalModel(heads, 64, C)
    else:
        ref, net = oracle.Transformer2DModel(heads, 64, C, cross_attention_dim=64), L.Transformer2DModel(heads, 64, C, cross_attention_dim=64)
    ref = ref.eval()
    state = seeded_state(ref)
    ref.load_state_dict(state)
    net.load_state_dict(state)
    net = net.half().eval()
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(g.tokens, C, generator=gen)
    text = torch.randn(g.clips * 7, 64, generator=gen)

    def run():
        ops.TRACE = []
        with torch.no_grad():
            y = net.tokens(x.half(), g) if temporal else net.tokens(x.half(), g, text.half(), 7)
        n, ops.TRACE = len(ops.TRACE), None
        return y.float(), n

    x5 = x.reshape(g.clips, g.frames, g.h, g.w, C)
    with torch.no_grad():
        if temporal:
            w = ref(x5.permute(0, 1, 4, 2, 3).reshape(g.images, C, g.h, g.w), num_frames=g.frames).sample
        else:
            w = ref(x5.permute(0, 1, 4, 2, 3).reshape(g.images, C, g.h, g.w),
                    encoder_hidden_states=text.reshape(g.clips, 7, 64).repeat_interleave(g.frames, 0)).sample
    w = w.permute(0, 2, 3, 1).reshape(g.tokens, C)
    assert L.LINEAR_ROWS
    tiles, n_tiles = run()                                     # (45 rows: below LINEAR_ROWS_MIN, the tile family)
    monkeypatch.setattr(L, "LINEAR_ROWS_MIN", 0)
    rows, n_rows = run()
    monkeypatch.setattr(L, "GN_FOLD", False)                   # the GroupNorm in front of proj_in as its own statistics + normalise pair
    unfolded, n_unfolded = run()
    assert n_unfolded == n_rows and rel_err(unfolded, rows) < 2e-3
    monkeypatch.setattr(L, "LINEAR_ROWS", False)
    off, n_off = run()
    assert n_off == n_tiles and torch.equal(off, tiles)
    assert n_tiles - n_rows == (3 if temporal else 5)
    assert rel_err(rows, w) < 2e-2 and rel_err(tiles, w) < 2e-2
    assert rel_err(rows, tiles) < 1e-2


@pytest.mark.parametrize("temporal", [False, True])
def test_feedforward_and_proj_out_as_one_kernel(emu, monkeypatch, temporal):
    """layers.FF_FUSED: norm3 -> GEGLU -> ff-out -> + x -> proj_out -> + residual of Transformer2DModel / TransformerTemporalModel at 320
    channels as ONE kernel (ops.ff_fused; oracle/layers.py:205-284): against the oracle module, against the two-contraction form, two
    contraction launches fewer per transformer (and no statistics epilogue on the producer), rebuilt when a weight changes in place."""
    from animate_anything_amd import layers as L, ops
    torch.manual_seed(3)
    C, heads, g = 320, 5, L.Grid(1, 3, 3, 5)
    if temporal:
        ref, net = oracle.TransformerTemporalModel(heads, 64, C), L.TransformerTemporalModel(heads, 64, C)
    else:
        ref, net = oracle.Transformer2DModel(heads, 64, C, cross_attention_dim=64), L.Transformer2DModel(heads, 64, C, cross_attention_dim=64)
    ref = ref.eval()
    state = seeded_state(ref)
    ref.load_state_dict(state)
    net.load_state_dict(state)
    net = net.half().eval()
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(g.tokens, C, generator=gen)
    text = torch.randn(g.clips * 7, 64, generator=gen)

    def run():
        ops.TRACE = []
        with torch.no_grad():
            y = net.tokens(x.half(), g) if temporal else net.tokens(x.half(), g, text.half(), 7)
        n, ops.TRACE = len(ops.TRACE), None
        return y.float(), n

    def want():
        x5 = x.reshape(g.clips, g.frames, g.h, g.w, C)
        with torch.no_grad():
            if temporal:
                y = ref(x5.permute(0, 1, 4, 2, 3).reshape(g.images, C, g.h, g.w), num_frames=g.frames).sample
            else:
                y = ref(x5.permute(0, 1, 4, 2, 3).reshape(g.images, C, g.h, g.w),
                        encoder_hidden_states=text.reshape(g.clips, 7, 64).repeat_interleave(g.frames, 0)).sample
        return y.permute(0, 2, 3, 1).reshape(g.tokens, C)

    assert L.FF_FUSED and net.fused_ff() is not None
    fused, n_fused = run()
    monkeypatch.setattr(L, "FF_FUSED", False)
    two, n_two = run()
    monkeypatch.setattr(L, "FF_FUSED", True)
    w = want()
    assert n_two - n_fused == 2                                # the GEGLU and the merged ff-out / proj_out contractions are gone
    assert rel_err(fused, w) < 2e-2 and rel_err(two, w) < 2e-2
    assert rel_err(fused, two) < 1e-2
    with torch.no_grad():                                      # an in-place edit of any of its weights rebuilds the stream
        net.proj_out.weight.mul_(0.5)
        net.transformer_blocks[0].ff.net[0].proj.bias.add_(0.25)
        net.transformer_blocks[0].norm3.weight.mul_(1.5)
        ref.load_state_dict({k: v.float() for k, v in net.state_dict().items()})
    again, _ = run()
    assert rel_err(again, want()) < 2e-2 and rel_err(again, fused) > 0.05
