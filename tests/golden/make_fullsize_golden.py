"""Golden output of the CPU oracle at the BENCHMARKED configuration (BASELINE.json configs[1]: full v1.02 architecture,
16 frames x 64x64 latents, CFG batch 2 => [2,4,16,64,64] in, 17 frames inside): one UNet3D forward, fp32, seeded
weights and inputs (tests/util.py `fullsize_oracle` / `fullsize_inputs`).  44 TFLOP on the host cores: minutes.

The fixture lets the `-m gpu` parity test at the metric configuration (tests/test_gpu_fullsize.py) run without
re-spending those minutes on every GPU box; the test recomputes the oracle live when the fixture is absent
(AA_FULLSIZE_LIVE=1 forces that).  Like tests/golden/oracle_golden.pt these are outputs of the ORACLE, not of the
reference ("parity unpinned").  Also records the wall time = the honest `cpu_baseline` of one
full-size step on this machine's cores (profiles/r02_cpu_baseline.json is written from it).

`--mask-image /root/reference/example/qingming2_label.jpg` (BASELINE.json configs[1] as written) replaces the synthetic centred
square by the reference's example mask taken through the reference's own mask path (train.py:750-764, restated in
`reference_mask_path` below); the latent-size mask is stored in the fixture (`mask`), which then travels to the GPU box
without /root/reference: unet_fullsize_16x64x64_qingming.pt.

Run from the repo root:  python tests/golden/make_fullsize_golden.py [--frames 16 --lat 64] [--mask-image PATH]
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from util import fullsize_inputs, fullsize_oracle  # noqa: E402


def reference_mask_path(path, height, width, h, w):
    """/root/reference/train.py:750-764, line by line: PIL open + resize to the pixel size, every non-zero value -> 255,
    T.ToTensor() (uint8 -> [0, 1] float, [1, H, W]), T.Resize([h, w], antialias=False) (torchvision on a tensor = bilinear
    F.interpolate, align_corners=False), 'b h w -> b 1 1 h w'.  Returns (np_mask uint8 [H, W], mask fp32 [1, 1, 1, h, w])."""
    import numpy as np
    import torch.nn.functional as F
    from PIL import Image
    mask = Image.open(path)                                     # :751
    mask = mask.resize((width, height))                         # :752
    np_mask = np.array(mask)                                    # :753
    np_mask[np_mask != 0] = 255                                 # :754
    t = torch.from_numpy(np_mask.astype("float32") / 255.0)[None]          # T.ToTensor(): [1, H, W]   (:761)
    t = F.interpolate(t[None], size=(h, w), mode="bilinear", align_corners=False, antialias=False)[0]   # T.Resize (:763)
    return np_mask, t[:, None, None]                            # rearrange 'b h w -> b 1 1 h w' (:764)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mask-image", default="", help="reference example mask (BASELINE configs[1]: example/qingming2_label.jpg)")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--lat", type=int, default=64)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--timing-json", default="", help="only time the forward and write the record here (no fixture): "
                    "the honest cpu_baseline of one full-size step on this host (profiles/r02_cpu_baseline.json)")
    a = ap.parse_args()
    cores = a.threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    ref, _ = fullsize_oracle()
    i = fullsize_inputs(a.frames, a.lat)
    extra, suffix = {}, ""
    if a.mask_image:
        _, m = reference_mask_path(a.mask_image, a.lat * 8, a.lat * 8, a.lat, a.lat)
        i["mask"] = m
        extra = {"mask": m.clone(), "mask_source": os.path.basename(a.mask_image), "mask_mean": m.mean().item()}
        suffix = "_" + os.path.splitext(os.path.basename(a.mask_image))[0].split("_")[0].rstrip("0123456789")
    t0 = time.perf_counter()
    with torch.no_grad():
        out = ref(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
    dt = time.perf_counter() - t0
    name = f"unet_fullsize_{a.frames}x{a.lat}x{a.lat}{suffix}"
    if not a.timing_json:
        torch.save({"out": out.half(), "abs_max": out.abs().max().item(), "seconds": dt, "cores": cores, **extra},
                   os.path.join(HERE, name + ".pt"))
    rec = {"config": f"full v1.02 UNet3D forward, CFG batch 2, {a.frames}+1 frames, {a.lat}x{a.lat} latents, fp32 oracle",
           "seconds_per_step": dt, "steps_per_s": 1.0 / dt, "cores": cores,
           "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") if os.path.exists("/proc/cpuinfo") else "?"}
    print(json.dumps(rec))
    with open(a.timing_json or os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
