"""Golden output of the CPU oracle at BASELINE.json configs[3] (the SVD path: full stable-video-diffusion-img2vid
architecture with the reference's 9 input channels, 14 frames x 72x128 latents = 576x1024 pixels, CFG batch 2): one
UNetSpatioTemporalConditionModel forward, fp32, seeded weights and inputs (tests/util.py `fullsize_svd_oracle` /
`svd_unet_inputs`).  The spatial self-attention of the oracle is evaluated one image at a time here (the batched score
tensor [28,5,9216,9216] fp32 would be 47 GB) - same arithmetic, bounded memory.

Like the other fixtures these are outputs of the ORACLE, not of the reference ("parity unpinned").
Run from the repo root:  python tests/golden/make_svd_golden.py [--frames 14 --h 72 --w 128]
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
from util import fullsize_svd_oracle, svd_unet_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--h", type=int, default=72)
    ap.add_argument("--w", type=int, default=128)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    cores = a.threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    import oracle.layers as L
    batched = L.Attention.forward

    def one_at_a_time(self, hidden_states, encoder_hidden_states=None):
        if hidden_states.dim() == 3 and hidden_states.shape[0] > 1 and hidden_states.shape[1] >= 4096:
            ctx = encoder_hidden_states
            return torch.cat([batched(self, hidden_states[i:i + 1], None if ctx is None else ctx[i:i + 1])
                              for i in range(hidden_states.shape[0])])
        return batched(self, hidden_states, encoder_hidden_states)

    L.Attention.forward = one_at_a_time
    ref, _ = fullsize_svd_oracle()
    i = svd_unet_inputs(2, a.frames, a.h, a.w)
    t0 = time.perf_counter()
    with torch.no_grad():
        out = ref(i["sample"], i["t"], i["text"], i["ids"]).sample
    dt = time.perf_counter() - t0
    name = f"svd_unet_fullsize_{a.frames}x{a.h}x{a.w}"
    torch.save({"out": out.half(), "abs_max": out.abs().max().item(), "seconds": dt, "cores": cores}, os.path.join(HERE, name + ".pt"))
    rec = {"config": f"full SVD UNetSpatioTemporalConditionModel forward (9 input channels), CFG batch 2, {a.frames} frames, "
                     f"{a.h}x{a.w} latents, fp32 oracle", "seconds_per_step": dt, "steps_per_s": 1.0 / dt, "cores": cores,
           "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") if os.path.exists("/proc/cpuinfo") else "?"}
    print(json.dumps(rec))
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
