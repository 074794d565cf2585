"""Builds libqbnn_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc: one object per translation unit
(compiled in parallel, rebuilt only when it or a header changed), then one link."""
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
UNITS = ["qbnn_kernels.hip", "qbnn_blocks.hip", "qbnn_down_ring.hip", "qbnn_chain_ring.hip", "qbnn_w16.hip", "qbnn_c48.hip", "qbnn_misc.hip", "qbnn_f32.hip", "qbnn_q8t.hip", "qbnn_small.hip"]
HEADERS = ["qbnn_conv.h", "qbnn_host.h", "qbnn_rng.h", "qbnn_common.h", "qbnn_eps_table.h", "qbnn_q8.h"]
SRC = [os.path.join(CSRC, u) for u in UNITS if os.path.exists(os.path.join(CSRC, u))]
DEPS = SRC + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(os.path.dirname(HERE), "include", "qbnn.h")]
LIB = os.environ.get("QBNN_LIB_OVERRIDE") or os.path.join(HERE, "libqbnn_hip.so")   # override: scratch ablation builds
OBJDIR = os.path.join(HERE, "build")                                                  # git-ignored

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value", "-std=c++17"]
EXTRA = os.environ.get("QBNN_HIPCC_EXTRA", "").split()                               # e.g. -DQBNN_STAMP for a diagnostic library


def kernel_source_sha16():
    """Hash of every kernel source: measurement files (profiles/*_pmc_traffic.json) are keyed to it."""
    h = hashlib.sha256()
    for p in sorted(SRC + [os.path.join(CSRC, x) for x in HEADERS]):
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in DEPS)


def _obj(src):
    tag = hashlib.sha256((" ".join(EXTRA) + LIB).encode()).hexdigest()[:8]
    return os.path.join(OBJDIR, os.path.basename(src) + "." + tag + ".o")


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_t = max(os.path.getmtime(p) for p in DEPS if p not in SRC)

    def compile_one(src):
        obj = _obj(src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            return obj
        cmd = [hipcc] + HIPCC_FLAGS + EXTRA + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SRC))) as ex:
        objs = list(ex.map(compile_one, SRC))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
