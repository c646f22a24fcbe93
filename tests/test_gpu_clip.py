"""CLIP text encoder on the GPU against `transformers.CLIPTextModel` itself (the real dependency, fp32 on the host): the
OpenCLIP ViT-H/14 text tower the reference's checkpoints carry (1024 wide, 23 layers, gelu) and the OpenAI ViT-L/14 one
(768 wide, 12 layers, quick_gelu), 77 tokens."""
import pytest
import torch

from animate_anything_amd.clip import CLIPTextModel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,cfg,dtype", [
    ("openclip-vit-h", dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16, hidden_act="gelu"), torch.float16),
    ("openai-vit-l", dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, hidden_act="quick_gelu"), torch.float16),
    ("openclip-vit-h", dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16, hidden_act="gelu"), torch.bfloat16),
])
def test_clip_text_model_matches_transformers(name, cfg, dtype):
    from transformers import CLIPTextConfig
    from transformers import CLIPTextModel as HFCLIPTextModel
    torch.manual_seed(0)
    cfg = dict(cfg, vocab_size=49408, max_position_embeddings=77, eos_token_id=2)
    ref = HFCLIPTextModel(CLIPTextConfig(**cfg, bos_token_id=0, pad_token_id=1)).eval()
    net = CLIPTextModel(**cfg).eval()
    net.load_state_dict({(k if k.startswith("text_model.") else "text_model." + k): v for k, v in ref.state_dict().items()})
    net = net.to("cuda", dtype)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1000, 40000, (2, 77), generator=g)
    ids[:, 0] = 49406
    ids[0, 9:], ids[1, 30:] = 49407, 49407                # end of text, then padded with it (what the tokenizer produces)
    with torch.no_grad():
        want = ref(ids).last_hidden_state
        got = net(ids.cuda())[0]
    err = ((got.float().cpu() - want).abs().max() / want.abs().max()).item()
    assert got.shape == (2, 77, cfg["hidden_size"]) and err < (2e-2 if dtype == torch.float16 else 1e-1), err
    mse = ((got.float().cpu() - want) ** 2).mean().item() / (want ** 2).mean().item()
    assert mse < (1e-4 if dtype == torch.float16 else 1e-2), mse


@pytest.mark.parametrize("name,cfg", [
    ("vit-h-14", dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, projection_dim=1024, hidden_act="gelu")),
    ("vit-l-14", dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, projection_dim=768, hidden_act="quick_gelu")),
])
def test_clip_vision_tower_matches_transformers(name, cfg):
    """The SVD path's image encoder (`CLIPVisionModelWithProjection`, ViT-H/14: 16 heads of 80 channels -> the vector-ALU
    attention kernel; ViT-L/14: heads of 64 -> the matrix-core kernel) against transformers itself, 224x224 input."""
    from transformers import CLIPVisionConfig
    from transformers import CLIPVisionModelWithProjection as HFVision
    from animate_anything_amd.clip import CLIPVisionModelWithProjection
    torch.manual_seed(0)
    cfg = dict(cfg, image_size=224, patch_size=14)
    ref = HFVision(CLIPVisionConfig(**cfg)).eval()
    net = CLIPVisionModelWithProjection(**cfg).eval()
    net.load_state_dict(ref.state_dict())
    net = net.to("cuda", torch.float16)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        want = ref(x).image_embeds
        got = net(x.cuda()).image_embeds
    err = ((got.float().cpu() - want).abs().max() / want.abs().max()).item()
    assert got.shape == (2, cfg["projection_dim"]) and err < 2e-2, err
