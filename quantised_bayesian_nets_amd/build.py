"""Builds libqbnn_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", "qbnn_kernels.hip"), os.path.join(HERE, "csrc", "qbnn_f32.hip"),
       os.path.join(HERE, "csrc", "qbnn_small.hip")]
DEPS = SRC + [os.path.join(HERE, "csrc", "qbnn_rng.cuh"), os.path.join(HERE, "csrc", "qbnn_common.h"), os.path.join(os.path.dirname(HERE), "include", "qbnn.h")]
LIB = os.environ.get("QBNN_LIB_OVERRIDE") or os.path.join(HERE, "libqbnn_hip.so")   # override: scratch ablation builds

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden",
               "-Wno-unused-value", "-std=c++17"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + SRC + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
