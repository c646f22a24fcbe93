"""Golden latents of the CPU oracle PIPELINE LOOP at the benchmarked configuration (BASELINE.json configs[1]: full v1.02
architecture, 16 frames x 64x64 latents, guidance 9 => CFG batch 2): the latents after EACH of the first `--steps` denoising
steps of the 25-step DPM-Solver++ schedule, fp32, seeded weights and inputs (tests/util.py `fullsize_oracle` /
`fullsize_inputs`, the same tensors as unet_fullsize_16x64x64.pt).  The loop is oracle.LatentToVideoPipeline.__call__ - the
restatement of /root/reference/models/pipeline.py:163-198 that tests/test_reference_pin.py holds equal, step by step, to the
reference's own `LatentToVideoPipeline.__call__`.

Why: the product's DEFAULT loop computes the text-independent prefix of the UNet once per guidance pair
(`LatentToVideoPipeline.cfg_shared_prefix`); round 4 only compared it with the product's own strict form.  With this fixture
`tests/test_gpu_fullsize.py::test_three_steps_against_the_oracle_at_the_metric_configuration` compares BOTH forms with the
oracle after every step (VERDICT r04 "next round" item 1).

One step is one 44 TFLOP forward on the host cores (225 s on the 8 build-container cores): three steps = ~12 minutes.
Like unet_fullsize_16x64x64.pt these are outputs of the ORACLE, not of the reference ("parity unpinned").

Run from the repo root:  python tests/golden/make_fullsize_multistep_golden.py [--steps 3]
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

GUIDANCE = 9.0          # example/train_mask_motion.yaml:137-147 (the reference's default sampling settings)
NUM_INFERENCE_STEPS = 25


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--lat", type=int, default=64)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    import oracle
    cores = a.threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    ref, _ = fullsize_oracle()
    i = fullsize_inputs(a.frames, a.lat)
    sched = oracle.DPMSolverMultistepScheduler()
    pipe = oracle.LatentToVideoPipeline(vae=None, unet=ref, scheduler=sched)
    sched.set_timesteps(NUM_INFERENCE_STEPS)
    ts = [int(t) for t in sched.timesteps][: a.steps]
    per_step, stamps = [], [time.perf_counter()]

    def keep(_i, _t, lat):
        per_step.append(lat.clone())
        stamps.append(time.perf_counter())
        print(f"step {_i} t={int(_t)} |x|max={lat.abs().max().item():.4f} {stamps[-1] - stamps[-2]:.1f} s", flush=True)

    # [neg; text] is how fullsize_inputs stacks the guidance pair; the pipeline concatenates them itself
    neg, text = i["text"][:1], i["text"][1:]
    pipe(height=a.lat * 8, width=a.lat * 8, num_frames=a.frames, num_inference_steps=NUM_INFERENCE_STEPS, guidance_scale=GUIDANCE,
         latents=i["sample"][:1].clone(), prompt_embeds=text, negative_prompt_embeds=neg, condition_latent=i["cond"][:1],
         mask=i["mask"], timesteps=ts, motion=[3.0], callback=keep, callback_steps=1, return_dict=False)
    name = f"pipeline_fullsize_{a.frames}x{a.lat}x{a.lat}_{a.steps}steps"
    torch.save({"latents": torch.stack(per_step), "timesteps": ts, "guidance_scale": GUIDANCE,
                "num_inference_steps": NUM_INFERENCE_STEPS, "motion": 3.0,
                "abs_max": [x.abs().max().item() for x in per_step]}, os.path.join(HERE, name + ".pt"))
    rec = {"config": f"oracle LatentToVideoPipeline loop, full v1.02 UNet3D, guidance {GUIDANCE}, {a.frames}+1 frames, "
                     f"{a.lat}x{a.lat} latents, first {a.steps} of {NUM_INFERENCE_STEPS} DPM-Solver++ steps, fp32",
           "timesteps": ts, "seconds_per_step": [stamps[k + 1] - stamps[k] for k in range(len(per_step))], "cores": cores}
    print(json.dumps(rec))
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
