"""Weight-file lookup of a diffusers / transformers checkpoint directory (host side)."""
import os

import torch


def load_state(root: str, stem: str, variant: str | None = None, bin_stem: str | None = None):
    """The state dict stored under `root`: `<stem>.<variant>.safetensors`, `<stem>.safetensors`, `<bin_stem>.<variant>.bin`,
    `<bin_stem>.bin` - first match wins (diffusers `variant="fp16"` checkpoints hold only the `.fp16.` files; the reference's SVD
    eval loads that way, train_svd.py:806-811)."""
    bin_stem = bin_stem or stem
    if variant is not None and (not isinstance(variant, str) or not variant.replace("-", "").replace("_", "").isalnum()):
        raise ValueError(f"checkpoint variant {variant!r}: a plain tag like 'fp16' (it is spliced into the file name)")
    tried = []
    for v in ((variant,) if variant else ()) + (None,):
        name = f"{stem}.{v}.safetensors" if v else f"{stem}.safetensors"
        path = os.path.join(root, name)
        tried.append(name)
        if os.path.exists(path):
            from safetensors.torch import load_file
            return load_file(path)
    for v in ((variant,) if variant else ()) + (None,):
        name = f"{bin_stem}.{v}.bin" if v else f"{bin_stem}.bin"
        path = os.path.join(root, name)
        tried.append(name)
        if os.path.exists(path):
            # pickled checkpoints: tensors only (a downloaded .bin must not get to run code); AA_UNSAFE_PICKLE=1 opts out for
            # checkpoints that carry other objects
            return torch.load(path, map_location="cpu", weights_only=os.environ.get("AA_UNSAFE_PICKLE", "0") != "1")
    raise FileNotFoundError(f"no weight file in {root} (tried {', '.join(tried)})")
