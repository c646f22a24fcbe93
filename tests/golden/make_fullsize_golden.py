"""Golden output of the CPU oracle at the BENCHMARKED configuration (BASELINE.json configs[1]: full v1.02 architecture,
16 frames x 64x64 latents, CFG batch 2 => [2,4,16,64,64] in, 17 frames inside): one UNet3D forward, fp32, seeded
weights and inputs (tests/util.py `fullsize_oracle` / `fullsize_inputs`).  44 TFLOP on the host cores: minutes.

The fixture lets the `-m gpu` parity test at the metric configuration (tests/test_gpu_fullsize.py) run without
re-spending those minutes on every GPU box; the test recomputes the oracle live when the fixture is absent
(AA_FULLSIZE_LIVE=1 forces that).  Like tests/golden/oracle_golden.pt these are outputs of the ORACLE, not of the
reference ("parity unpinned").  Also records the wall time = the honest `cpu_baseline` of one
full-size step on this machine's cores (profiles/r02_cpu_baseline.json is written from it).

Run from the repo root:  python tests/golden/make_fullsize_golden.py [--frames 16 --lat 64]
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


def main():
    ap = argparse.ArgumentParser()
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
    t0 = time.perf_counter()
    with torch.no_grad():
        out = ref(i["sample"], i["t"], i["text"], i["cond"], i["mask"], motion=i["motion"]).sample
    dt = time.perf_counter() - t0
    name = f"unet_fullsize_{a.frames}x{a.lat}x{a.lat}"
    if not a.timing_json:
        torch.save({"out": out.half(), "abs_max": out.abs().max().item(), "seconds": dt, "cores": cores},
                   os.path.join(HERE, name + ".pt"))
    rec = {"config": f"full v1.02 UNet3D forward, CFG batch 2, {a.frames}+1 frames, {a.lat}x{a.lat} latents, fp32 oracle",
           "seconds_per_step": dt, "steps_per_s": 1.0 / dt, "cores": cores,
           "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") if os.path.exists("/proc/cpuinfo") else "?"}
    print(json.dumps(rec))
    with open(a.timing_json or os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
