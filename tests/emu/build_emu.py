"""Build tests/emu/libaa_emu.so: the C ABI compiled for the host on the SIMT emulator (test infra)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build(force=False):
    out = os.path.join(HERE, "libaa_emu.so")
    srcs = [os.path.join(HERE, "aa_api_emu.cpp")]
    deps = srcs + [os.path.join(HERE, "dev.h")]
    csrc = os.path.join(ROOT, "animate_anything_amd", "csrc")
    for d, _, fs in os.walk(csrc):
        deps += [os.path.join(d, f) for f in fs if f.endswith((".h", ".hip"))]
    deps.append(os.path.join(ROOT, "include", "aa_mi355.h"))
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    cmd = [CLANG, "-O2", "-mf16c", "-mavx2", "-mfma", "-std=c++17", "-shared", "-fPIC", "-pthread", "-Wno-unused-value", "-Wno-psabi",
           "-I", HERE, "-I", os.path.join(csrc, "kernels"), "-I", csrc, "-I", os.path.join(ROOT, "include"),
           "-o", out] + srcs
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force=True))
