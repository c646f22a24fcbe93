"""Host-side pieces of the eval driver (config merge, image / mask preprocessing) - CPU only."""
import numpy as np
import torch
import yaml
from PIL import Image

from animate_anything_amd import eval as aa_eval


def test_config_dotlist_merge(tmp_path):
    yaml.safe_dump({"pretrained_model_path": "x", "validation_data": {"num_frames": 16, "prompt": "a"}},
                   open(tmp_path / "c.yaml", "w"))
    cfg = aa_eval.load_config(str(tmp_path / "c.yaml"), ["validation_data.num_frames=8", "validation_data.mask=m.jpg", "seed=3"])
    assert cfg.validation_data.num_frames == 8 and cfg.validation_data.mask == "m.jpg" and cfg.seed == 3
    assert "mask" in cfg.validation_data and cfg.validation_data.get("strength", 5) == 5
    cfg.validation_data.height = 440                      # eval() mutates the node in place (train.py:743-744)
    assert cfg["validation_data"]["height"] == 440


def test_image_and_mask_preprocessing():
    img = Image.fromarray(np.full((60, 80, 3), 255, dtype=np.uint8))
    x = aa_eval.preprocess_image(img, 64, 96)
    assert x.shape == (1, 3, 64, 96) and torch.allclose(x, torch.ones_like(x))
    m = np.zeros((64, 96), dtype=np.uint8)
    m[:, 48:] = 255
    lat = aa_eval.mask_to_latent(m, 8, 12)
    assert lat.shape == (1, 1, 1, 8, 12) and lat[0, 0, 0, 0, 0] == 0 and lat[0, 0, 0, 0, -1] == 1
