"""Builds mrfa_amd/_lib/libmrfa_hip.so (all HIP kernels + the C ABI of include/mrfa_hip.h) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting .so travels to the
GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libmrfa_hip.so")
SOURCES = ["error.cpp", "conv_mfma.hip", "conv_split.hip", "conv_halo.hip", "conv_small.hip", "conv_lean.hip", "wgrad_mfma.hip", "wgrad_split.hip", "wgrad_halo.hip", "wgrad_small.hip", "wgrad_lean.hip", "conv_fewout.hip", "conv_fewout3.hip", "layout.hip", "pack_multi.hip", "norm.hip", "sample.hip", "elementwise.hip", "optim.hip", "tokenpose.hip", "attention_mfma.hip", "losses.hip", "prior.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wno-unused-result"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(ROOT, "include", "mrfa_hip.h"), os.path.join(CSRC, "common.h")]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            raise FileNotFoundError(sp)
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            cmd = [hipcc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", obj]
            if verbose:
                print("[mrfa_amd.build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[mrfa_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
