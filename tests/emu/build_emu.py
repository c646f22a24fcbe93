"""Build tests/emu/libaa_emu.so: the C ABI compiled for the host on the SIMT emulator (test infra)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
TU_GROUPS = 8          # aa_api_impl.h: AA_TU_GROUPS (the explicit instantiations listed there)


def build(force=False):
    out = os.path.join(HERE, "libaa_emu.so")
    srcs = [os.path.join(HERE, "aa_api_emu.cpp")]
    deps = srcs + [os.path.join(HERE, "dev.h"), os.path.join(HERE, "aa_tiles_emu.cpp")]
    csrc = os.path.join(ROOT, "animate_anything_amd", "csrc")
    for d, _, fs in os.walk(csrc):
        deps += [os.path.join(d, f) for f in fs if f.endswith((".h", ".hip"))]
    deps.append(os.path.join(ROOT, "include", "aa_mi355.h"))
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    # the C ABI unit + TU_GROUPS units with a slice of the contraction tile table each (aa_api_impl.h: AA_TU_GROUPS), compiled in
    # parallel like the HIP build
    flags = ["-O2", "-mf16c", "-mavx2", "-mfma", "-std=c++17", "-fPIC", "-pthread", "-Wno-unused-value", "-Wno-psabi",
             f"-DAA_TU_GROUPS={TU_GROUPS}", "-I", HERE, "-I", os.path.join(csrc, "kernels"), "-I", csrc, "-I", os.path.join(ROOT, "include")]
    obj_dir = os.path.join(HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    jobs = [([srcs[0]], os.path.join(obj_dir, "aa_api_emu.o"))]
    jobs += [([f"-DAA_TU_GROUP={g}", os.path.join(HERE, "aa_tiles_emu.cpp")], os.path.join(obj_dir, f"aa_tiles_emu_{g}.o")) for g in range(TU_GROUPS)]

    def compile_one(job):
        src, obj = job
        subprocess.check_call([CLANG] + flags + ["-c"] + src + ["-o", obj])
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
        objs = list(pool.map(compile_one, jobs))
    subprocess.check_call([CLANG, "-shared", "-fPIC", "-pthread"] + objs + ["-o", out])
    return out


if __name__ == "__main__":
    print(build(force=True))
