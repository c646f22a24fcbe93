"""Build libaa_mi355.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libaa_mi355.so")


def _deps():
    deps = [os.path.join(ROOT, "include", "aa_mi355.h")]
    for d, _, fs in os.walk(CSRC):
        deps += [os.path.join(d, f) for f in fs if f.endswith((".h", ".hip"))]
    return deps


PROBE_LIB = os.path.join(PKG, "libaa_mi355_probe.so")
TU_GROUPS = 8             # aa_api_impl.h: tile-table entry i is compiled in unit i % TU_GROUPS


def build(force=False, verbose=False, probe=False, ablate=0):
    """Compile csrc/aa_api.hip + 8 x csrc/aa_tiles.hip -> libaa_mi355.so (skipped when up to date). Returns the path.
    probe=True builds the -DAA_PHASE_PROBE profiling variant (scripts/phase_probe.py) next to it;
    ablate=bits a timing-only -DAA_X_ABLATE variant of the hand-scheduled kernels (scripts/x_ablate.py)."""
    LIB = PROBE_LIB if probe else globals()["LIB"]
    if ablate:
        LIB = os.path.join(PKG, f"libaa_mi355_abl{ablate}.so")
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in _deps()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", f"-DAA_TU_GROUPS={TU_GROUPS}",
             "-I", os.path.join(CSRC, "kernels", "device"), "-I", os.path.join(CSRC, "kernels"), "-I", CSRC,
             "-I", os.path.join(ROOT, "include")]
    if probe:
        flags.insert(0, "-DAA_PHASE_PROBE")
    if ablate:
        flags.insert(0, f"-DAA_X_ABLATE={int(ablate)}")
    if verbose:
        flags.insert(0, "-Rpass-analysis=kernel-resource-usage")
    # one translation unit for the C ABI + TU_GROUPS units with a slice of the contraction tile table each, compiled in parallel
    # (a single unit took more than five minutes), linked into one shared object
    obj_dir = os.path.join(PKG, "build", os.path.basename(LIB)[:-3])
    os.makedirs(obj_dir, exist_ok=True)
    jobs = [([os.path.join(CSRC, "aa_api.hip")], os.path.join(obj_dir, "aa_api.o"))]
    jobs += [([f"-DAA_TU_GROUP={g}", os.path.join(CSRC, "aa_tiles.hip")], os.path.join(obj_dir, f"aa_tiles_{g}.o")) for g in range(TU_GROUPS)]

    def compile_one(job):
        src, obj = job
        subprocess.check_call([hipcc] + flags + ["-c"] + src + ["-o", obj])
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
        objs = list(pool.map(compile_one, jobs))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose="-v" in sys.argv, probe="--probe" in sys.argv))
