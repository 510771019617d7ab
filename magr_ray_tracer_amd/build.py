"""Build the native libraries in-tree (hipcc for gfx950, g++/gcc for the host side).

    python -m magr_ray_tracer_amd.build [--force]

Outputs (git-ignored, shipped to the GPU box by gpurun):
    magr_ray_tracer_amd/librt355.so       device path: HIP kernels + C-ABI (include/rt355.h)
    magr_ray_tracer_amd/librt355_refb.so  the same with the reference's OpenCL builtin sequences (tests only, -DRT355_REF_BUILTINS)
    magr_ray_tracer_amd/librt355_host.so  host side: Scene / BVH2 / BVH4 / TLAS / Renderer mirror
    oracle/liboracle.so                   CPU restatement (test infrastructure only)
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "magr_ray_tracer_amd")
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# -ffp-contract=off + correctly rounded div/sqrt: the float discipline of oracle/oracle.c
DEVICE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-pthread"]
ORACLE_FLAGS = ["-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-fopenmp", "-Wall", "-D_GNU_SOURCE"]


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_device(force=False):
    src = os.path.join(PKG, "csrc", "rt355.hip")
    deps = [src, os.path.join(PKG, "csrc", "rt355_kernels.h"), os.path.join(ROOT, "include", "rt355.h"),
            os.path.join(ROOT, "include", "rt355_types.h")]
    out = os.path.join(PKG, "librt355.so")
    if force or _stale(out, deps):
        _run([HIPCC] + DEVICE_FLAGS + [src, "-o", out])
    return out


def build_device_refb(force=False):
    """librt355_refb.so: the same library with -DRT355_REF_BUILTINS (normalize / length / exp / sin / cos / acospi / atan2pi as ROCm's
    OpenCL library evaluates them for the reference's kernels, rt355_kernels.h).  Test infrastructure for tests/test_gpu_reference.py
    (uncurated whole-frame comparison with the reference's kernels); the shipped library is librt355.so."""
    src = os.path.join(PKG, "csrc", "rt355.hip")
    deps = [src, os.path.join(PKG, "csrc", "rt355_kernels.h"), os.path.join(ROOT, "include", "rt355.h"),
            os.path.join(ROOT, "include", "rt355_types.h")]
    out = os.path.join(PKG, "librt355_refb.so")
    if force or _stale(out, deps):
        # -Bsymbolic: this library defines the same global symbols as librt355.so (C-ABI entry points, the kernels' host stubs).  Loaded
        # into a process that already holds librt355.so, its own references would otherwise bind to THAT library's definitions - and
        # launch the other build's kernels
        _run([HIPCC] + DEVICE_FLAGS + ["-DRT355_REF_BUILTINS", "-Wl,-Bsymbolic", src, "-o", out])
    return out


def build_host(force=False):
    hdir = os.path.join(PKG, "host")
    srcs = [os.path.join(hdir, f) for f in sorted(os.listdir(hdir)) if f.endswith(".cpp")]
    deps = srcs + [os.path.join(hdir, "rt_host.h"), os.path.join(ROOT, "include", "rt355.h"),
                   os.path.join(ROOT, "include", "rt355_host.h"), os.path.join(ROOT, "include", "rt355_types.h")]
    out = os.path.join(PKG, "librt355_host.so")
    if force or _stale(out, deps):
        _run(["g++"] + HOST_FLAGS + srcs + ["-o", out, "-L" + PKG, "-lrt355", "-lz", "-Wl,-rpath,$ORIGIN"])
    return out


def build_examples(force=False):
    """examples/headless_tick: the reference's main loop, headless, on the host mirror + C-ABI (C++ drop-in check)."""
    src = os.path.join(ROOT, "examples", "headless_tick.cpp")
    out = os.path.join(ROOT, "examples", "headless_tick")
    deps = [src, os.path.join(PKG, "host", "rt_host.h"), os.path.join(PKG, "librt355_host.so"), os.path.join(PKG, "librt355.so")]
    if force or _stale(out, deps):
        _run(["g++", "-O2", "-std=c++17", "-Wall", src, "-o", out, "-L" + PKG, "-lrt355_host", "-lrt355", "-Wl,-rpath,$ORIGIN/../magr_ray_tracer_amd"])
    return out


def build_oracle(force=False):
    odir = os.path.join(ROOT, "oracle")
    src = os.path.join(odir, "oracle.c")
    out = os.path.join(odir, "liboracle.so")
    if force or _stale(out, [src, os.path.join(odir, "oracle.h"), os.path.join(ROOT, "include", "rt355_types.h")]):
        _run(["gcc"] + ORACLE_FLAGS + [src, "-o", out, "-lm"])
    return out


def build_ref(force=False):
    """oracle/_ref: the reference's own OpenCL kernels compiled for gfx950 (only where /root/reference exists)."""
    script = os.path.join(ROOT, "oracle", "build_ref.sh")
    if os.path.isdir("/root/reference") and os.path.exists(script):
        _run(["bash", script] + (["--force"] if force else []))


def build_all(force=False):
    build_device(force)
    build_device_refb(force)
    build_host(force)
    build_examples(force)
    build_oracle(force)
    build_ref(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
