"""The reference run a second time on ANOTHER CPU code path, for the fixture generators (runs only in the build container).

A fixture's fp32 outputs are compared at 1e-5 relative; what absolute floor goes with that is measured, not chosen: the generator
re-runs ITSELF in a child process with `--alt-out FILE` under ALT_ENV (ATen and MKL restricted to their AVX2 kernels instead of
AVX-512: other vector widths, other summation orders in sgemm / the vectorised pointwise kernels; conv graphs also switch oneDNN off
inside the generator) and records how far the reference is from itself (`refspread.*` in the fixture).  The tests bound the build's
deviation by a small multiple of that spread, stated beside each assertion: twice for the conv graphs (whose two reference runs
also differ in the conv backend, oneDNN against plain ATen), four times for the MLPs (two runs of the same sgemm on two vector widths:
the spread of ONE pair of summation orders, and the build's order is a third one)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ALT_ENV = {"ATEN_CPU_CAPABILITY": "avx2", "MKL_ENABLE_INSTRUCTIONS": "AVX2", "ONEDNN_MAX_CPU_ISA": "AVX2"}


def alt_out_path():
    """The path given as `--alt-out PATH` (this process IS the alternative run), or None."""
    if "--alt-out" in sys.argv:
        return sys.argv[sys.argv.index("--alt-out") + 1]
    return None


def run_alt(script, tag):
    """Runs `script --alt-out tmp --alt-tag tag` under ALT_ENV and returns the arrays it saved."""
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "alt.npz")
        env = dict(os.environ, **ALT_ENV)
        subprocess.run([sys.executable, script, "--alt-out", path, "--alt-tag", tag], env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        d = np.load(path)
        return {k: d[k] for k in d.files}


def alt_tag():
    return sys.argv[sys.argv.index("--alt-tag") + 1] if "--alt-tag" in sys.argv else None


def spread(a, b, floor=1e-30):
    """(max abs, max rel) distance of two runs of the reference."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max()), float((np.abs(a - b) / np.maximum(np.minimum(np.abs(a), np.abs(b)), floor)).max())
