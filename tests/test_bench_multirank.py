"""bench.py's N > 1 control flow on CPU: `python bench.py --gpus 2` respawns itself under torch.distributed.run, the two ranks
check WORLD_SIZE, own one clip each (different seeds), run the real denoising loop, gather the final latents inside the timed
region, take the max over ranks and rank 0 prints ONE JSON line.  tests/bench_selftest.py calls bench.main() with RCCL swapped for
gloo, the GPU library for the test suite's CPU build of the same kernels and the 1.4 G-parameter architecture for a toy one -
nothing else; the line says so.  bench.py itself refuses to run on a machine without GPUs."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=600, script="tests/bench_selftest.py"):
    # the ranks load the suite's CPU build of the kernels: bring it up to date HERE, once - two ranks rebuilding a stale library at the same
    # time race on the file (one rank dlopens while the other relinks)
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build()
    env = dict(os.environ, AA_EMU_THREADS="2", OMP_NUM_THREADS="2", **env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


def test_two_ranks_through_the_script_itself():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 0 only, exactly one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert "selftest" in out and "NOT a measurement" in out["selftest"]
    assert out["config"]["clips"] == 2 and out["config"]["parallelism"] == "clip-sharded x2"
    assert len(out["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in out["per_rank_ms_per_step"])
    # whole-job value = clips of all ranks per second of the SLOWEST rank
    assert abs(out["value"] - 2 * 1e3 / max(out["per_rank_ms_per_step"])) / out["value"] < 0.02
    assert out["other_form"]["cfg_shared_prefix"] is True and out["other_form"]["value"] > 0


def test_world_size_mismatch_is_refused():
    env = {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_itself_has_no_cpu_path():
    if torch.cuda.is_available():
        return
    assert "tests" not in [ln.split()[-1].strip('"\',)') for ln in open(os.path.join(ROOT, "bench.py")) if "sys.path" in ln]
    r = _run(["--steps", "1", "--warmup", "0"], {}, script="bench.py")
    assert r.returncode != 0 and "no CPU fallback" in r.stderr
