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


def build(force=False, verbose=False, probe=False, ablate=0):
    """Compile csrc/aa_api.hip -> libaa_mi355.so (skipped when up to date). Returns the path.
    probe=True builds the -DAA_PHASE_PROBE profiling variant (scripts/phase_probe.py) next to it;
    ablate=bits a timing-only -DAA_X_ABLATE variant of the hand-scheduled kernels (scripts/x_ablate.py)."""
    LIB = PROBE_LIB if probe else globals()["LIB"]
    if ablate:
        LIB = os.path.join(PKG, f"libaa_mi355_abl{ablate}.so")
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in _deps()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffast-math",
           "-I", os.path.join(CSRC, "kernels", "device"), "-I", os.path.join(CSRC, "kernels"), "-I", CSRC,
           "-I", os.path.join(ROOT, "include"),
           os.path.join(CSRC, "aa_api.hip"), "-o", LIB]
    if probe:
        cmd.insert(1, "-DAA_PHASE_PROBE")
    if ablate:
        cmd.insert(1, f"-DAA_X_ABLATE={int(ablate)}")
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose="-v" in sys.argv, probe="--probe" in sys.argv))
