"""CPU-side tests (no GPU): host logic, C-ABI surface, weight packing, sample sharding, and the world_size-2
gloo path of the MC reduction (the oracle stands in for the per-rank GPU evaluation)."""
import ctypes as C
import os
import re
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from quantised_bayesian_nets_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "qbnn.h")).read()
    declared = set(re.findall(r"\b(qbnn_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"qbnn_sample_params", "qbnn_conv_desc", "qbnn_head_desc", "qbnn_block_desc"}
    L = _lib.lib()
    for sym in sorted(declared):
        assert hasattr(L, sym), sym
    assert set(_lib.EXPORTS) == declared
    # the loaded library, the header and the ctypes binding agree on the ABI version (a changed prototype bumps all three)
    assert L.qbnn_version() == _lib.ABI_VERSION == int(re.search(r"#define QBNN_ABI_VERSION (\d+)", hdr).group(1))


def test_errors_without_gpu_are_loud(golden_w8):
    import quantised_bayesian_nets_amd as q
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args).load_reference_state(golden_w8["state"])
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32))          # CPU tensor: the product has no CPU fallback
    with pytest.raises(NotImplementedError):
        q.ModelFactory.get_model("conv_lenet", [1, 1, 28, 28], 10, True, args)
    bad = types.SimpleNamespace(activation_precision=8, weight_precision=8)
    with pytest.raises(AssertionError):       # reference quant_utils.py:120: activations are at most 7 bit
        q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, bad)


def test_pack_weights_layout_and_ones_row():
    """QBNN_LAYOUT_MFMA32: k rows padded to 32 bytes, lane-fragment order, all-ones row for ragged cout."""
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for cout, kh, krow in [(24, 3, 72), (48, 3, 144), (96, 1, 48), (192, 3, 576), (24, 1, 27)]:
        k = kh * krow
        w = rng.integers(-128, 128, (cout, k)).astype(np.int8)
        n = L.qbnn_packed_weight_bytes(cout, k, krow, 0)
        rbp = (krow + 31) // 32 * 32
        KS, NT = kh * rbp // 32, (cout + 31) // 32
        assert n == NT * KS * 1024
        dst = np.full(n, 77, np.int8)
        _lib.check(L.qbnn_pack_weights_host(w.ctypes.data_as(C.c_void_p), cout, k, krow, 0, dst.ctypes.data_as(C.c_void_p)))
        d = dst.reshape(NT, KS, 64, 16)
        for (nn, kk) in [(0, 0), (cout - 1, k - 1), (cout // 2, min(krow, k - 1)), (5, krow - 1)]:
            kp = (kk // krow) * rbp + kk % krow
            assert d[nn // 32, kp // 32, ((kp // 16) % 2) * 32 + nn % 32, kp % 16] == w[nn, kk]
        # pads are zero; ragged cout carries the ones row at n == cout
        total_ones = int((dst == 1).sum()) - int((w == 1).sum())
        if cout % 32:
            ones = d[cout // 32, :, :, :].reshape(KS, 2, 32, 16)[:, :, cout % 32, :].reshape(-1)   # [KS*32] in kp order
            valid = np.array([(kp % rbp) < krow for kp in range(KS * 32)])
            assert np.array_equal(ones == 1, valid) and total_ones == valid.sum()
        assert int((dst != 0).sum()) <= cout * k + (k if cout % 32 else 0)
    # row-major layout is a plain copy
    w = rng.integers(-128, 128, (10, 192)).astype(np.int8)
    dst = np.zeros(L.qbnn_packed_weight_bytes(10, 192, 192, 1), np.int8)
    _lib.check(L.qbnn_pack_weights_host(w.ctypes.data_as(C.c_void_p), 10, 192, 192, 1, dst.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(dst[:1920].reshape(10, 192), w)


def test_reference_state_roundtrip(golden_w8):
    import quantised_bayesian_nets_amd as q
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args).load_reference_state(golden_w8["state"])
    assert m.n_weights() == 1571592 and len(m.stochastic_layers()) == 21
    for name, layer in zip(m.stochastic_layer_names(), m.stochastic_layers()):
        st = layer.reference_state(name + ".")
        for k, v in st.items():
            assert np.array_equal(np.asarray(v), np.asarray(golden_w8["state"][k])), k
    assert [l.layer_id for l in m.stochastic_layers()] == list(range(21))


def test_shard_samples_partitions_the_range():
    from quantised_bayesian_nets_amd.mc import shard_samples
    for S in (1, 7, 100, 1024):
        for G in (1, 2, 3, 8):
            parts = [shard_samples(S, r, G) for r in range(G)]
            assert sum(c for _, c in parts) == S
            pos = 0
            for b, c in parts:
                assert b == pos
                pos += c
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_finalize_moments_matches_torch():
    from quantised_bayesian_nets_amd.mc import finalize_moments
    p = torch.rand(9, 5, 10)
    p64 = p.double()
    mom = torch.stack([p64.sum(0), (p64 * p64).sum(0)])                  # the device kernel keeps the sums in fp64
    mean, var = finalize_moments(mom, 9)
    torch.testing.assert_close(mean, p.mean(0), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(var, p.var(0), rtol=1e-5, atol=1e-8)     # unbiased, as experiments/utils.py:352
    # small spread against the mean (what made fp32 sums cancel): still 1e-5 from fp64 sums
    q_ = 0.9 + 1e-4 * torch.rand(100, 7, 3)
    q64 = q_.double()
    mean, var = finalize_moments(torch.stack([q64.sum(0), (q64 * q64).sum(0)]), 100)
    torch.testing.assert_close(var, q64.var(0).float(), rtol=1e-5, atol=0)


_WORKER = r"""
import os, sys, types, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from conftest import load_golden
from oracle import oracle as orc
from quantised_bayesian_nets_amd.mc import shard_samples, all_reduce_moments, finalize_moments
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = load_golden("resnet_bbb_a7w8.npz")
net = orc.Int8ResNetOracle(g["state"], 7, 8)
S, seed = 5, 3
begin, count = shard_samples(S, rank, world)
x = g["x"][:2]
mom = torch.zeros(2, 2, 10, dtype=torch.float64)
for s in range(begin, begin + count):          # the oracle stands in for this rank's GPU evaluation
    p = torch.from_numpy(net.forward(x, seed, s)).double()
    mom[0] += p; mom[1] += p * p
all_reduce_moments(mom)
mean, var = finalize_moments(mom, S)
if rank == 0:
    _, ps = net.mc_predict(x, S, seed)
    ps = torch.from_numpy(ps)
    torch.testing.assert_close(mean, ps.mean(0), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(var, ps.double().var(0).float(), rtol=1e-5, atol=1e-9)
    # samples 0..2 of the global stream are the golden ones recorded from the reference
    torch.testing.assert_close(ps[:3], torch.from_numpy(g["probs"][:, :2]), rtol=1e-5, atol=1e-8)
    print("OK")
# a QAT model with live observers is sequential in the sample index: mc_predict must refuse to shard it
from quantised_bayesian_nets_amd.mc import mc_predict
class _Seq:
    sequential_samples = True
try:
    mc_predict(_Seq(), torch.zeros(1), 4, 0)
    raise SystemExit("sharded a sequential model")
except RuntimeError as e:
    assert "cannot be sharded" in str(e)
dist.destroy_process_group()
"""


def test_world_size_2_gloo_reduce_equals_single_process(tmp_path):
    """Sharding S over 2 ranks + one sum all-reduce of the [2,B,C] moments == the single-process MC loop."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", OMP_NUM_THREADS="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0]


def test_model_factory_routes_and_loads_reference_state_without_a_gpu():
    """Host logic only (no kernel launch): ModelFactory keeps the reference's names / switches (q, args.qat_eval), the
    float / QAT / MC-Dropout graphs load the reference-format state dicts of the fixtures, and a CPU tensor is refused."""
    import types
    import numpy as np
    import pytest
    import torch
    import quantised_bayesian_nets_amd as q
    from conftest import load_golden

    def st(name):
        d = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
        return {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}

    a = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=8, p=0.1)
    f = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, False, a).load_reference_state(st("resnet_bbb_f32.npz"))
    assert type(f).__module__.endswith("models_f32") and len(f.stochastic_named()) == 21
    assert [m.layer_id for _, m in f.stochastic_named()] == list(range(21))
    aq = types.SimpleNamespace(**vars(a), qat_eval=True)
    m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, aq).load_reference_state(st("resnet_bbb_qat.npz"))
    assert type(m).__module__.endswith("models_qat")
    mn, mx = m.layers[0].weight_fake_quant.min_max()
    assert np.isfinite(mn) and np.isfinite(mx) and mn < 0 < mx
    c = m.layers[0].scale_factor()                         # conv_qat.py:140-141
    assert c.shape == (24,) and bool((c > 0).all())
    mc = q.ModelFactory.get_model("conv_resnet_mc", [1, 3, 32, 32], 10, True, a).load_reference_state(load_golden("resnet_mc_a7w8.npz")["state"])
    assert len(mc.dropouts()) == 20 and [d.layer_id for d in mc.dropouts()] == list(range(20))
    le = q.ModelFactory.get_model("conv_lenet_bbb", [1, 28, 28], 10, False, a).load_reference_state(st("lenet_bbb_f32.npz"))
    with pytest.raises(RuntimeError):
        le.forward_mc(torch.zeros(2, 1, 28, 28))           # no CPU fallback
    with pytest.raises(NotImplementedError):
        q.ModelFactory.get_model("no_such_model", [1, 3, 32, 32], 10, True, a)


def test_public_header_is_valid_c99():
    """include/qbnn.h is the C ABI: it must compile as plain C (a maintainer binds it from C / cgo / ctypes generators)."""
    hdr = os.path.join(ROOT, "include", "qbnn.h")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # every struct the Python binding mirrors has the same size as the C one
    from quantised_bayesian_nets_amd import _lib
    prog = '#include <stdio.h>\n#include "qbnn.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n", sizeof(qbnn_sample_params), sizeof(qbnn_conv_desc), sizeof(qbnn_block_desc), sizeof(qbnn_down_desc), sizeof(qbnn_head_desc));return 0;}\n'
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "s.c")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(td, "s")])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(td, "s")]).split()]
    assert sizes == [C.sizeof(_lib.SampleParams), C.sizeof(_lib.ConvDesc), C.sizeof(_lib.BlockDesc), C.sizeof(_lib.DownDesc), C.sizeof(_lib.HeadDesc)]


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus 2` with no launcher spawns 2 fresh ranks under torch.distributed.run as a child and relays
    rank 0's JSON line (here: the no-GPU plumbing mode over gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-check"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["plumbing_check"] is True and out["ranks"] == 2 and out["backend"] == "gloo" and "rccl_ranks" not in out and out["rank_sum"] == 1.0 and out["moments_sum"] == 3.0


def test_bench_self_launches_eight_ranks_with_the_baseline_partitions():
    """The driver's N = 8 case, rehearsed without GPUs: `python bench.py --gpus 8` spawns 8 fresh ranks (gloo), every rank takes its
    contiguous block of BASELINE configs[4] (1024 MC samples -> 8 x 128, global Philox sample indices) and of configs[3] (16 ensemble
    members -> 8 x 2), the fp64 moments are sum-all-reduced, and rank 0's one JSON line names the 8 ranks and the backend it ran on (gloo: no RCCL claim)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plumbing-check"], env=env,
                       capture_output=True, text=True, timeout=480)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["ranks"] == 8 and out["backend"] == "gloo" and "rccl_ranks" not in out and out["rank_sum"] == 28.0 and out["moments_sum"] == 36.0
    by_rank = sorted(out["shards"], key=lambda d: d["rank"])
    assert [d["samples_1024"] for d in by_rank] == [[128 * r, 128] for r in range(8)]
    assert [d["members_16"] for d in by_rank] == [[2 * r, 2] for r in range(8)]
    # the one driver command at N = 8 also measures BASELINE configs[4] and configs[3] (bench.py: secondary_multi), partitioned the same way
    sec = {d["workload"]: d for d in out["secondary_multi"]}
    assert sec["resnet_bbb_w4"]["units_global"] == 1024 and sec["resnet_bbb_w4"]["shards"] == [[128 * r, 128] for r in range(8)]
    assert sec["ensemble16"]["units_global"] == 16 and sec["ensemble16"]["shards"] == [[2 * r, 2] for r in range(8)]


def test_n24_packed_layout_places_every_weight_and_a_ones_row_per_tile():
    """QBNN_LAYOUT_MFMA32_N24 (include/qbnn.h, round 5): 24 output channels + a ones row per fragment tile.  Host packing only (no GPU):
    every logical weight sits at tile n // 24, row n % 24 of the layout's (ks, k-half, byte) position, row 24 of EVERY tile is 1 at the
    valid k positions, rows 25..31 and the kernel-row pads are 0 -- and the MFMA32 form of the same weights differs only in the tiling."""
    import ctypes as C
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    cout, cin, ksz = 48, 48, 3
    k, krow = ksz * ksz * cin, ksz * cin
    w = rng.integers(-128, 128, (cout, k), dtype=np.int8)
    nb = L.qbnn_packed_weight_bytes(cout, k, krow, 2)
    rbp = (krow + 31) // 32 * 32
    KS = ksz * rbp // 32
    assert nb == 2 * KS * 1024 == L.qbnn_packed_weight_bytes(cout, k, krow, 0)
    out = np.zeros(nb, np.int8)
    _lib.check(L.qbnn_pack_weights_host(w.ctypes.data_as(C.c_void_p), cout, k, krow, 2, out.ctypes.data_as(C.c_void_p)))
    frag = out.reshape(2, KS, 2, 32, 16)          # [tile][k-step][k-half][row][byte]
    dense = frag.transpose(0, 3, 1, 2, 4).reshape(2, 32, KS * 32)      # [tile][row][padded k]
    kp = (np.arange(k) // krow) * rbp + np.arange(k) % krow
    for n in range(cout):
        assert np.array_equal(dense[n // 24, n % 24, kp], w[n]), n
    valid = np.zeros(KS * 32, bool)
    valid[kp] = True
    for t in range(2):
        assert np.array_equal(dense[t, 24], valid.astype(np.int8))
        assert not dense[t, 25:].any() and not dense[t, :24][:, ~valid].any()
    assert L.qbnn_packed_weight_bytes(40, k, krow, 2) == 0            # cout % 24 != 0: not an N24 shape
    assert L.qbnn_pack_weights_host(w.ctypes.data_as(C.c_void_p), 40, k, krow, 2, out.ctypes.data_as(C.c_void_p)) != 0


def test_tail_packed_layout_gathers_the_kernel_rows_ragged_ends():
    """QBNN_LAYOUT_MFMA32_TAIL (include/qbnn.h, round 5) on a 24 -> 24 3x3 conv: kernel rows of 72 bytes = two full k-steps + an 8-byte tail;
    the three tails share the seventh k-step (7 KiB per conv instead of 9).  Host packing only: every weight at its place, the ones row,
    zeros elsewhere; shapes the layout is not defined for are refused."""
    import ctypes as C
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(6)
    cout, k, krow = 24, 216, 72
    w = rng.integers(-128, 128, (cout, k), dtype=np.int8)
    assert L.qbnn_packed_weight_bytes(cout, k, krow, 3) == 7 * 1024 and L.qbnn_packed_weight_bytes(cout, k, krow, 0) == 9 * 1024
    out = np.zeros(7 * 1024, np.int8)
    _lib.check(L.qbnn_pack_weights_host(w.ctypes.data_as(C.c_void_p), cout, k, krow, 3, out.ctypes.data_as(C.c_void_p)))
    dense = out.reshape(7, 2, 32, 16).transpose(2, 0, 1, 3).reshape(32, 7 * 32)      # [row][packed k]
    kk = np.arange(k)
    kh, j = kk // krow, kk % krow
    kp = np.where(j < 64, kh * 64 + j, 192 + kh * 8 + (j - 64))
    assert len(set(kp.tolist())) == k
    assert np.array_equal(dense[:cout][:, kp], w)
    valid = np.zeros(7 * 32, bool); valid[kp] = True
    assert np.array_equal(dense[cout], valid.astype(np.int8)) and not dense[cout + 1:].any() and not dense[:cout][:, ~valid].any()
    assert L.qbnn_packed_weight_bytes(48, 432, 144, 3) == 0              # 16-byte tails x 3 rows do not fit one k-step
    assert L.qbnn_packed_weight_bytes(96, 864, 288, 3) == 0              # no ragged end at all


def test_ring_kernels_issue_no_flat_loads(tmp_path):
    """The LDS-DMA weight rings (csrc/qbnn_chain_ring.hip, qbnn_down_ring.hip) keep their own `s_waitcnt vmcnt` accounting: a wave waits for ITS
    share of a slab by counting the vector-memory instructions it issued after it (advisor, round 4).  That holds for global_load / global_load_lds,
    which return in order -- a FLAT load (what a pointer the compiler cannot prove global turns into) returns out of order and would let the MFMAs
    read a half-landed slab.  Disassemble both units for gfx950 and check: no flat_load in any ring kernel, and the ring's DMA is there."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    n_kernels = 0
    for unit in ("qbnn_chain_ring.hip", "qbnn_down_ring.hip"):
        out = tmp_path / (unit + ".s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-Wno-unused-value", "-I" + os.path.join(root, "include"),
                               "--cuda-device-only", "-S", "-o", str(out), os.path.join(root, "quantised_bayesian_nets_amd", "csrc", unit)],
                              stderr=subprocess.DEVNULL)
        parts = re.split(r"\n(_ZN?\w+):", out.read_text())
        for name, body in zip(parts[1::2], parts[2::2]):
            if "ring" not in name or "_kernel" not in name:
                continue
            n_kernels += 1
            assert not re.search(r"^\s+flat_load", body, re.M), name
            assert "global_load_lds_dwordx4" in body, name
    assert n_kernels >= 12


def test_design_table_is_what_the_generator_prints():
    """DESIGN.md section 4.3's per-kernel table is generated from profiles/r06_*.json (tools/design_table.py --write): the committed block must be
    what the generator prints from the committed measurement files, so the prose cannot drift from the counters (verdict round 5, item 6)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_table.py"), "r06", "r05"], capture_output=True, text=True, check=True).stdout.strip()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    a, b = design.index("<!-- GENERATED: tools/design_table.py -->"), design.index("<!-- END GENERATED -->") + len("<!-- END GENERATED -->")
    assert design[a:b].strip() == out


def test_qat_conv_grid_contract_without_a_gpu():
    """qbnn_conv2d_q8_blocks is host logic: the number of (min, max) partial pairs per sample the chosen conv form will write, which the caller sizes the
    observer's buffer by.  The ResNet's 3 x 3 geometries and its 3-channel stem take the LDS-tiled forms of csrc/qbnn_q8t.hip (blocks of whole output rows /
    images x channel groups); anything else -- other map sizes, 1 x 1 convs, Cout % 4 != 0 -- the gather forms' 64 x 64 / 128 x 32 tiles."""
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    B = 256
    tiled = {(32, 3, 24, 1): B * 8 // 8, (32, 24, 24, 1): B * 8 // 8, (32, 24, 48, 2): B * 2 // 2, (16, 48, 48, 1): B * 2 // 2, (16, 48, 96, 2): B // 2,
             (8, 96, 96, 1): B // 2, (8, 96, 192, 2): (B // 8) * 2, (4, 192, 192, 1): (B // 8) * 2, (32, 24, 40, 1): (B * 8 // 8) * 2, (8, 96, 100, 1): (B // 2) * 2}
    for (H, cin, cout, stride), want in tiled.items():
        assert L.qbnn_conv2d_q8_blocks(B, H, H, cin, cout, 3, stride, 1) == want, (H, cin, cout, stride)
    assert L.qbnn_conv2d_q8_blocks(3, 8, 8, 96, 96, 3, 1, 1) == 2                      # a ragged image group still gets its block
    gather = lambda npix, cout, narrow: ((npix + 127) // 128) * ((cout + 31) // 32) if narrow else ((npix + 63) // 64) * ((cout + 63) // 64)
    assert L.qbnn_conv2d_q8_blocks(B, 16, 16, 24, 24, 3, 1, 1) == gather(B * 256, 24, True)       # a 16 x 16 map at 24 channels: no tiled form
    assert L.qbnn_conv2d_q8_blocks(B, 16, 16, 48, 96, 1, 2, 0) == gather(B * 64, 96, False)       # the 1 x 1 shortcut
    assert L.qbnn_conv2d_q8_blocks(B, 32, 32, 24, 22, 3, 1, 1) == gather(B * 1024, 22, True)      # Cout % 4 != 0
    assert L.qbnn_add_q8_blocks(16) == 1 and L.qbnn_add_q8_blocks(256 * 32 * 32 * 24) == 512
