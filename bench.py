#!/usr/bin/env python3
"""Benchmark of the hot path: MC forward samples/sec, ResNet-18 BBB int8 (A7/W8), batch 256 (BASELINE.json).

One step = one pass of the Monte-Carlo evaluation over one batch: S stochastic forwards of the same 256 CIFAR-shaped
images (weight sampling + 20 int8 convs + head per sample) reduced to predictive moments.  With N GPUs every rank
evaluates `--samples` samples (weak scaling: global S = N * samples, Philox subsequence = global sample index) and the
[2,B,C] partial moments are summed with one RCCL all-reduce per step.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--samples S] [--batch B]
  N > 1: either launched by torch.distributed.run (one rank per GPU: RANK / LOCAL_RANK / WORLD_SIZE in the environment), or
  started plainly -- then this process touches no GPU, spawns `python -m torch.distributed.run --nproc-per-node N bench.py ...`
  as a child and relays rank 0's JSON line and the child's exit code.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the one with the largest share of GPU time in
the timed region), measured with HIP events on the launch stream; `cpu_baseline` times the CPU oracle (a port of the
reference algorithm, bit-identical to it on the golden vectors) on this host's cores over a bounded sample.
"""
import argparse
import json
import re
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_I8_PEAK_TOPS = 5000.0   # int8 dense = 2x bf16 (2.5 PF)
VALU_OPS_PLAIN, VALU_OPS_RES = 6, 13     # epilogue vector instructions per output element in the shipped ISA (profiles/r02_isa_epilogue.txt)
VALU_OPS_PLAIN_MAGIC, VALU_OPS_RES_MAGIC = 5, 12      # ... where the accumulators start at 1.5 * 2^23 (round 5: the layer-1 kernel; one v_sub_f32 for sub + cvt)
# What a SIMD issues of that instruction mix, measured IN the kernel clock domain (tools/issue_bench.hip, profiles/r03_issue_bench.txt:
# s_memtime cycles, every SIMD busy, the epilogue's own dependent sequence sub / cvt / fma / mul / min / cvt_pk): 4.09 cycles per
# wave-instruction with 2 resident waves per SIMD, 2.92 with 4, 2.21 with 8 (one wave alone: 8.2; "fast" add / mul / fma / sub 3.0 /
# 1.9 / 1.4-1.6, "slow" min / med3 / cvt* / perm 4.5-5.3 / 3.4-3.5 / 1.7-2.3).  Round 2's table (3.1 / 6.8 "at 2.4 GHz") assumed a
# clock the chip does not hold and its cost-weighted roof is dropped; the roof below is issue-slot time at the kernel's occupancy.
VALU_ISSUE_CYCLES = {2: 4.09, 4: 2.92, 8: 2.21}
SHADER_CLOCK_HZ = 2.4e9                   # GRBM_GUI_ACTIVE / 8 / duration of the conv kernels: 2.37-2.43 GHz (profiles/r02_pmc_util.json)
MFMA_I8_SUSTAINED_TOPS = 3260.0   # measured: pure v_mfma_i32_32x32x32_i8 loop on random int8 operands, whole chip
                                  # (tools/mfma_sustained.hip, profiles/r01_mfma_sustained.txt; 4700 on all-zero operands)


def conv_algorithmic_bytes(S, B, H, Cin, Cout, ks, stride, nweights):
    """SURVEY.md 8(d) byte model: each conv reads its input and writes its output once (1 B/element) and reads its
    parameters (mu_q, sigma_q: 2 B/weight) once per sample."""
    Ho = H // stride
    return S * (B * (H * H * Cin + Ho * Ho * Cout) + 2 * nweights)


def conv_ops(S, B, H, Cin, Cout, ks, stride):
    Ho = H // stride
    return 2 * S * B * Ho * Ho * Cout * Cin * ks * ks


def kernel_source_sha16():
    from quantised_bayesian_nets_amd import build as _b
    return _b.kernel_source_sha16()


def _wl_resnet(w_bits, samples_default, tag):
    def build(a, world, q, load_golden):
        wb = w_bits or a.w_bits
        g = load_golden(f"resnet_bbb_a7w{wb}.npz")   # random-init, calibrated, converted int8 conv_resnet_bbb (recorded from the reference)
        args = types.SimpleNamespace(activation_precision=7, weight_precision=wb)
        a.w_bits = wb
        model = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args).load_reference_state(g["state"])
        S = a.samples if a.samples > 0 else samples_default
        x_host = torch.randn(a.batch, 3, 32, 32, generator=torch.Generator().manual_seed(2))     # synthetic CIFAR-shaped, normalised
        return dict(golden=g, model=model, x_host=x_host, units_per_gpu=S, units_global=S * world, resnet=True, cpu_baseline=True,
                    step=lambda m, x, S_, seed: q.mc_predict(m, x, S_, seed, return_var=True), scaling="weak", dtype="int8",
                    metric="MC forward samples/sec, ResNet-18 BBB int8 batch=256", unit="MC samples/s",
                    describe="%s: CIFAR-10-shaped ResNet-18 (24/48/96/192) Bayes-by-backprop, A7/W%d int8, %d MC samples per GPU per step, "
                             "batch=%d" % (tag, wb, S, a.batch))
    return build


def _wl_ensemble16(a, world, q, load_golden):
    """BASELINE configs[3]: 16 SGHMC members (deterministic int8 ResNets), members = the MC samples, sharded over the ranks (strong scaling)."""
    from fixtures import synth_ensemble_members, load_ensemble_fixture
    ge = load_ensemble_fixture()
    n = 16
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, model="conv_resnet_sgld", samples=n)
    model = q.ModelFactory.get_model("conv_resnet_sgld", [1, 3, 32, 32], 10, True, args, training_mode=False).load_reference_state(synth_ensemble_members(ge, n))
    x_host = torch.randn(a.batch, 3, 32, 32, generator=torch.Generator().manual_seed(2))
    return dict(golden=None, model=model, x_host=x_host, units_per_gpu=n // world, units_global=n, resnet=False, cpu_baseline=False,
                step=lambda m, x, S_, seed: q.mc_predict(m, x, S_, 0), scaling="strong", dtype="int8",
                metric="ensemble member forwards/sec, ResNet-18 SGHMC 16-member int8 batch=256", unit="member forwards/s",
                describe="configs[3]: CIFAR-10-shaped ResNet-18 SGHMC ensemble, 16 deterministic int8 members (2 recorded from the reference + 14 "
                         "deterministic perturbations), batch=%d, members sharded over the ranks" % a.batch)


def _wl_lenet_mc(a, world, q, load_golden):
    """BASELINE configs[1]: MNIST-shaped LeNet MC-Dropout A7/W8, 100 MC samples, batch 128."""
    g = load_golden("lenet_mc_a7w8.npz")
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=0.2)
    model = q.ModelFactory.get_model("conv_lenet_mc", [1, 1, 28, 28], 10, True, args).load_reference_state(g["state"])
    S = a.samples if a.samples > 0 else 100
    B = 128 if a.batch == 256 else a.batch
    x_host = torch.rand(B, 1, 28, 28, generator=torch.Generator().manual_seed(2))
    return dict(golden=g, model=model, x_host=x_host, units_per_gpu=S, units_global=S * world, resnet=False, cpu_baseline=False,
                step=lambda m, x, S_, seed: q.mc_predict(m, x, S_, seed), scaling="weak", dtype="int8", graph={},
                metric="MC forward samples/sec, LeNet MC-Dropout int8 batch=128", unit="MC samples/s",
                describe="configs[1]: MNIST-shaped LeNet MC-Dropout (p=0.2), A7/W8 int8, %d MC samples per GPU per step, batch=%d" % (S, B))


def _wl_lenet_bbb(a, world, q, load_golden):
    """SURVEY 8 row a6 `conv_lenet_bbb`: MNIST-shaped LeNet, int8 Bayes-by-backprop (sampled weights in every layer), 100 MC samples, batch 128."""
    g = load_golden("lenet_bbb_a7w8.npz")
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    model = q.ModelFactory.get_model("conv_lenet_bbb", [1, 1, 28, 28], 10, True, args).load_reference_state(g["state"])
    S = a.samples if a.samples > 0 else 100
    B = 128 if a.batch == 256 else a.batch
    x_host = torch.rand(B, 1, 28, 28, generator=torch.Generator().manual_seed(2))
    return dict(golden=g, model=model, x_host=x_host, units_per_gpu=S, units_global=S * world, resnet=False, cpu_baseline=False,
                step=lambda m, x, S_, seed: q.mc_predict(m, x, S_, seed), scaling="weak", dtype="int8",
                metric="MC forward samples/sec, LeNet BBB int8 batch=%d" % B, unit="MC samples/s",
                describe="conv_lenet_bbb: MNIST-shaped LeNet Bayes-by-backprop, A7/W8 int8, %d MC samples per GPU per step, batch=%d" % (S, B))


def _wl_resnet_mc(a, world, q, load_golden):
    """SURVEY 8 row a6 `conv_resnet_mc`: ResNet-18 with a quantised channel dropout behind every conv, deterministic int8 weights."""
    g = load_golden("resnet_mc_a7w8.npz")
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=g["meta"]["p"])
    model = q.ModelFactory.get_model("conv_resnet_mc", [1, 3, 32, 32], 10, True, args).load_reference_state(g["state"])
    S = a.samples if a.samples > 0 else 100
    x_host = torch.randn(a.batch, 3, 32, 32, generator=torch.Generator().manual_seed(2))
    return dict(golden=g, model=model, x_host=x_host, units_per_gpu=S, units_global=S * world, resnet=False, cpu_baseline=False,
                step=lambda m, x, S_, seed: q.mc_predict(m, x, S_, seed), scaling="weak", dtype="int8",
                metric="MC forward samples/sec, ResNet-18 MC-Dropout int8 batch=%d" % a.batch, unit="MC samples/s",
                describe="conv_resnet_mc: CIFAR-10-shaped ResNet-18 MC-Dropout (p=%.2f), A7/W8 int8, %d MC samples per GPU per step, batch=%d" % (g["meta"]["p"], S, a.batch))


def _wl_mlp_f32(a, world, q, load_golden):
    """BASELINE configs[0]: UCI-regression-shaped 4x100 MLP, Bayes-by-backprop fp32, 10 MC samples, 1000 rows."""
    d = np.load(os.path.join(ROOT, "tests", "golden", "mlp_bbb_f32.npz"))
    st = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    in_dim = int(d["meta.in_dim"])
    model = q.ModelFactory.get_model("linear_bbb", [in_dim], 1, False, types.SimpleNamespace(sigma_prior=-2.0)).load_reference_state(st)
    S = a.samples if a.samples > 0 else 10
    x_host = torch.randn(1000, in_dim, generator=torch.Generator().manual_seed(2))
    return dict(golden=None, model=model, x_host=x_host, units_per_gpu=S, units_global=S * world, resnet=False, cpu_baseline=False,
                step=lambda m, x, S_, seed: q.mc_predict_regression(m, x, S_, seed), scaling="weak", dtype="f32", graph=dict(regression=True),
                metric="MC forward samples/sec, 4x100 MLP BBB fp32, 1000 rows", unit="MC samples/s",
                describe="configs[0]: UCI-regression-shaped (in_dim %d) 4x100 MLP Bayes-by-backprop fp32, %d MC samples per GPU per step, 1000 rows" % (in_dim, S))


def _wl_mlp_bbb(a, world, q, load_golden):
    """SURVEY 8 row a6 `linear_bbb` with q=True: the UCI-regression-shaped MLP in its converted int8 form (sampled int8 weights), 10 MC samples, 1000 rows."""
    d = np.load(os.path.join(ROOT, "tests", "golden", "mlp_bbb_a7w8.npz"), allow_pickle=True)
    st = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    model = q.ModelFactory.get_model("linear_bbb", [13], 1, True, args).load_reference_state(st)
    S = a.samples if a.samples > 0 else 10
    x_host = torch.randn(1000, 13, generator=torch.Generator().manual_seed(2))
    return dict(golden=None, model=model, x_host=x_host, units_per_gpu=S, units_global=S * world, resnet=False, cpu_baseline=False,
                step=lambda m, x, S_, seed: q.mc_predict_regression(m, x, S_, seed), scaling="weak", dtype="int8", graph=dict(regression=True),
                metric="MC forward samples/sec, 4x100 MLP BBB int8, 1000 rows", unit="MC samples/s",
                describe="linear_bbb (q=True): UCI-regression-shaped (in_dim 13) 4x100 MLP Bayes-by-backprop, A7/W8 int8, %d MC samples per GPU per step, 1000 rows" % S)


def _wl_mlp_mc(a, world, q, load_golden):
    """SURVEY 8 row a6+ `linear_mc` (mcdropout/models_mc.py:10-73): the MC-Dropout regression MLP in its converted int8 form, 10 MC samples, 1000 rows."""
    d = np.load(os.path.join(ROOT, "tests", "golden", "mlp_mc_a7w8.npz"))
    st = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    in_dim, p = int(d["meta.in_dim"]), float(d["meta.p"])
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=p)
    model = q.ModelFactory.get_model("linear_mc", [in_dim], 1, True, args).load_reference_state(st)
    S = a.samples if a.samples > 0 else 10
    x_host = torch.randn(1000, in_dim, generator=torch.Generator().manual_seed(2))
    return dict(golden=None, model=model, x_host=x_host, units_per_gpu=S, units_global=S * world, resnet=False, cpu_baseline=False,
                step=lambda m, x, S_, seed: q.mc_predict_regression(m, x, S_, seed), scaling="weak", dtype="int8", graph=dict(regression=True),
                metric="MC forward samples/sec, 4x100 MLP MC-Dropout int8, 1000 rows", unit="MC samples/s",
                describe="linear_mc (q=True): UCI-regression-shaped (in_dim %d) 4x100 MLP MC-Dropout (p=%.2f), A7/W8 int8, %d MC samples per GPU per step, 1000 rows" % (in_dim, p, S))


def _wl_resnet_mc_f32(a, world, q, load_golden):
    """SURVEY 8 rows a6+ / a7 with q=False: the float MC-Dropout ResNet-18 (FloatFunctional BernoulliDropout behind every conv, dropout.py:15-40)."""
    d = np.load(os.path.join(ROOT, "tests", "golden", "resnet_mc_f32.npz"))
    st = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    p = float(d["meta.p"])
    model = q.ModelFactory.get_model("conv_resnet_mc", [1, 3, 32, 32], 10, False, types.SimpleNamespace(p=p)).load_reference_state(st)
    S = a.samples if a.samples > 0 else 10
    x_host = torch.randn(a.batch, 3, 32, 32, generator=torch.Generator().manual_seed(2))
    return dict(golden=None, model=model, x_host=x_host, units_per_gpu=S, units_global=S * world, resnet=False, cpu_baseline=False,
                step=lambda m, x, S_, seed: q.mc_predict(m, x, S_, seed), scaling="weak", dtype="f32",
                metric="MC forward samples/sec, ResNet-18 MC-Dropout fp32 batch=%d" % a.batch, unit="MC samples/s",
                describe="conv_resnet_mc (q=False): CIFAR-10-shaped ResNet-18 MC-Dropout (p=%.2f), fp32, %d MC samples per GPU per step, batch=%d" % (p, S, a.batch))


def _wl_resnet_float(kind):
    """SURVEY 8a rows a1 / a2 at the headline's shape (B = 256): the float Bayes-by-backprop ResNet-18 (fp32 MFMA convs, per-sample
    weights) and its QAT form evaluated with live observers (fake-quantised tensors; the convs as exact integer sums on the int8 matrix pipe since round 5,
    fp64 sums with QBNN_QAT_I8=0; the MC samples are sequential
    through the observers' EMA state, so 10 per step).  States recorded from the reference (tests/golden/resnet_bbb_{f32,qat}.npz)."""
    def build(a, world, q, load_golden):
        if kind == "qat":
            g = load_golden("resnet_bbb_qat.npz")
            args = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=8, qat_eval=True)
            model = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args).load_reference_state(g["state"])
        else:
            g = load_golden("resnet_bbb_f32.npz")
            model = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, False, types.SimpleNamespace(sigma_prior=-2.0)).load_reference_state(g["state"])
        S = a.samples if a.samples > 0 else 10
        x_host = torch.randn(a.batch, 3, 32, 32, generator=torch.Generator().manual_seed(2))
        what = "QAT fake-quant evaluation with live observers (A7/W8 grids)" if kind == "qat" else "float Bayes-by-backprop"
        # (the QAT pass is ~100 launches of 5 - 130 us: replayed as ONE captured HIP graph the row does not depend on the host's launch rate;
        #  bit-identical to the eager pass: tests/test_gpu_float_qat.py::test_qat_resnet_graph_replay_equals_eager.  --no-graph launches eagerly.)
        extra = dict(graph={}) if kind == "qat" and world == 1 else {}
        return dict(golden=None, model=model, x_host=x_host, units_per_gpu=S, units_global=S * world, resnet=False, cpu_baseline=False, **extra,
                    step=lambda m, x, S_, seed: q.mc_predict(m, x, S_, seed), scaling="weak", dtype=("int8+f32" if os.environ.get("QBNN_QAT_I8", "1") != "0" else "f64") if kind == "qat" else "f32",
                    metric="MC forward samples/sec, ResNet-18 BBB %s batch=%d" % ("QAT-eval" if kind == "qat" else "fp32", a.batch), unit="MC samples/s",
                    describe="rows a1/a2: CIFAR-10-shaped ResNet-18 (24/48/96/192), %s, %d MC samples per GPU per step, batch=%d" % (what, S, a.batch))
    return build


WORKLOADS = {"resnet_bbb": _wl_resnet(0, 100, "configs[2]"), "resnet_f32": _wl_resnet_float("f32"), "resnet_qat": _wl_resnet_float("qat"), "resnet_bbb_w4": _wl_resnet(4, 128, "configs[4] (A7/W4, 1024 samples over 8 GPUs = 128 per GPU)"),
             "ensemble16": _wl_ensemble16, "lenet_mc": _wl_lenet_mc, "lenet_bbb": _wl_lenet_bbb, "mlp_bbb": _wl_mlp_bbb, "mlp_f32": _wl_mlp_f32, "resnet_mc": _wl_resnet_mc,
             "mlp_mc": _wl_mlp_mc, "resnet_mc_f32": _wl_resnet_mc_f32}


def usable_cpus():
    """Threads the CPU leg may really run on: the smallest of the scheduler affinity, the cgroup CPU quota (a GPU box gives a one-GPU job a
    share of its host, not the 256 hardware threads os.cpu_count() reports) and the physical core count (SMT siblings add nothing to an
    int8 GEMM).  Returns (threads, how)."""
    how = {}
    how["affinity"] = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        import psutil
        how["physical"] = psutil.cpu_count(logical=False) or how["affinity"]
    except Exception:                        # noqa: BLE001
        how["physical"] = how["affinity"]
    quota = None
    try:
        q_, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q_ != "max":
            quota = max(1, int(float(q_) / float(per)))
    except Exception:                        # noqa: BLE001
        try:
            q_, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q_ > 0:
                quota = max(1, q_ // per)
        except Exception:                    # noqa: BLE001
            pass
    if quota:
        how["cgroup_quota"] = quota
    return max(1, min(how.values())), how


def cpu_baseline(a, g, x_host, seed):
    """The reference's CPU path timed on this host, on a bounded sample of the same workload (same batch, a few MC samples).
    Primary leg, kind "port", engine "torch-fbgemm" (oracle/fbgemm_baseline.py): the op sequence of the reference's int8 layers through PyTorch's own quantised CPU
    operators (ATen + FBGEMM) -- oracle/fbgemm_baseline.py, which reproduces the golden vectors recorded from the reference.
    Second leg, kind "port": the plain-C restatement oracle/qbnn_oracle.c (OpenMP), the parity checker.
    Round 6: the thread count is PINNED to the cores this process may use (usable_cpus: affinity, cgroup quota, physical cores) -- torch's
    default of one thread per hardware thread of the host oversubscribed the box's share and the single-sample times swung 0.6 - 5.6 s
    inside one run --, two warm-up samples, then 9 single-sample timings; value = 1 / median, the minimum beside it."""
    from oracle import oracle as orc
    from oracle.fbgemm_baseline import FbgemmResNetBBB
    threads, how = usable_cpus()
    if a.cpu_threads:
        threads = a.cpu_threads
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    xn = x_host.numpy()
    fb = FbgemmResNetBBB(g["state"], 7, a.w_bits)
    for _ in range(2):                                     # warm-up (thread pool, allocator, fbgemm's packing caches)
        fb.forward(xn)
    n_fb = a.cpu_samples or 9
    ts = []
    for _ in range(n_fb):
        t = time.perf_counter()
        fb.forward(xn)
        ts.append(time.perf_counter() - t)
    med = sorted(ts)[len(ts) // 2]
    torch.set_num_threads(prev_threads)
    net = orc.Int8ResNetOracle(g["state"], 7, a.w_bits)
    orc.lib().qbo_set_num_threads(threads)
    t = time.perf_counter()
    p_or = net.forward(xn, seed, 0)
    one = time.perf_counter() - t
    n_cpu = a.cpu_samples or max(1, min(8, int(10.0 / max(one, 1e-3))))
    t = time.perf_counter()
    for s in range(1, 1 + n_cpu):
        net.forward(xn, seed, s)
    el = time.perf_counter() - t
    return {"value": round(1.0 / med, 4), "unit": "MC samples/s", "cores": threads, "kind": "port", "engine": "torch-fbgemm",
            "sample": f"median of {n_fb} single MC samples of the same batch ({a.batch} images) through torch's quantised CPU ops "
                      f"(fbgemm engine; per layer normal_ -> quantize_per_tensor -> quantized.mul/add -> clamp -> conv2d_prepack -> "
                      f"quantized.conv2d(_relu) -> clamp), after 2 warm-up samples, torch.set_num_threads({threads}) = the cores this process may use",
            "value_best": round(1.0 / min(ts), 4),
            "seconds_per_sample_min_med_max": [round(min(ts), 4), round(med, 4), round(max(ts), 4)],
            "host_cpus": os.cpu_count(), "usable_cpus": how,
            "port": {"value": round(n_cpu / el, 4), "unit": "MC samples/s", "cores": orc.lib().qbo_num_threads(), "kind": "port", "engine": "plain C (oracle/qbnn_oracle.c, OpenMP)",
                     "sample": f"{n_cpu} MC samples of the same batch through oracle/qbnn_oracle.c (OpenMP), after 1 warm-up sample"},
            "_p_oracle_sample0": p_or}


# algorithmic work per MC sample at the workload's own batch (ops = 2 MAC; layer-granular bytes: each conv / linear reads its input and
# writes its output once).  ResNet-18 24/48/96/192: 78,522,240 MAC and 482,506 activation elements per image; LeNet: 6,522,000 MAC per image.
RESNET_MAC_PER_IMG, RESNET_ELTS_PER_IMG, LENET_MAC_PER_IMG = 78522240, 482506, 6522000
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 MFMA (v_mfma_f32_32x32x2f32), dense


def secondary_roofline(name, units_per_s, batch):
    """The whole-workload roofline view of a secondary row (no per-kernel HIP events here: one number per workload): achieved = algorithmic ops (or
    bytes) per MC sample x samples/s against the pipe that bounds it.  None for the launch-bound MLP workloads."""
    if name in ("resnet_f32", "resnet_mc_f32"):
        a = units_per_s * 2.0 * RESNET_MAC_PER_IMG * batch / 1e12
        return {"bound": "mfma", "achieved": round(a, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(a / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                "note": "fp32 matrix pipe (v_mfma_f32_32x32x2f32); channel padding (24 -> 32, 48 -> 64) leaves ~125 TFLOP/s of useful work at full issue"}
    if name == "resnet_qat":
        a = units_per_s * 2.0 * RESNET_MAC_PER_IMG * batch / 1e12
        gb = units_per_s * (RESNET_ELTS_PER_IMG * batch * (4 + 4 + 1) + 2 * 1571592 * 4) / 1e9      # every conv output: fp32 written, read by its fake-quantiser, int8 written
        return {"bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gb / HBM_PEAK_GBS, 4), "traffic": None,
                "mfma_achieved_tops": round(a, 2), "mfma_frac_of_i8_peak": round(a / MFMA_I8_PEAK_TOPS, 5),
                "note": "live observers force two passes per tensor (conv -> fp32 Z with its min / max, then the fake-quantiser): 9 B per activation element; "
                        "the integer sums themselves use a few per cent of the int8 pipe"}
    if name in ("resnet_bbb_w4", "ensemble16", "resnet_mc"):
        a = units_per_s * 2.0 * RESNET_MAC_PER_IMG * batch / 1e12
        return {"bound": "mfma", "achieved": round(a, 1), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s", "frac": round(a / MFMA_I8_PEAK_TOPS, 4), "traffic": None}
    if name in ("lenet_mc", "lenet_bbb"):
        a = units_per_s * 2.0 * LENET_MAC_PER_IMG * batch / 1e12
        return {"bound": "mfma", "achieved": round(a, 2), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s", "frac": round(a / MFMA_I8_PEAK_TOPS, 5), "traffic": None,
                "note": "launch-bound by design at this size (a 100-sample step is a fraction of a millisecond)"}
    return None


SECONDARY = ("resnet_bbb_w4", "ensemble16", "lenet_mc", "lenet_bbb", "mlp_f32", "mlp_bbb", "mlp_mc", "resnet_mc", "resnet_f32", "resnet_qat", "resnet_mc_f32")


def secondary_workloads(a, q, load_golden, seed):
    """BASELINE.json's other configs measured in the same run (N = 1, after the headline's timed region): 6 setup steps, then
    at least 10 timed steps (as many as fill ~60 ms for the sub-millisecond workloads) between synchronisations, on the same code paths `--workload NAME` runs.  Failures are reported, not hidden."""
    out = {}
    for name in SECONDARY:
        try:
            b = argparse.Namespace(**vars(a))
            b.workload, b.samples, b.batch, b.w_bits = name, 0, 256, 8
            wl = WORKLOADS[name](b, 1, q, load_golden)
            model, x = wl["model"], wl["x_host"].cuda()
            S = wl["units_global"]
            graphed = q.GraphedPredictor(model, S, **wl["graph"]) if ("graph" in wl and not a.no_graph) else None
            run = (lambda: graphed(x, seed)) if graphed is not None else (lambda: wl["step"](model, x, S, seed))
            slow = name in ("resnet_f32", "resnet_qat", "resnet_mc_f32")       # 6-16 ms per step
            for _ in range(6):                                    # (round 3 gave the slow ones 2 + 4 steps: a cold clock, -17 % against their own --workload runs)
                run()
            torch.cuda.synchronize()
            n = 10
            if not slow:                                          # sub-millisecond steps: time >= ~60 ms of them (clock ramp, host jitter)
                t0 = time.perf_counter()
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                n = max(10, min(200, int(0.06 / max((time.perf_counter() - t0) / 3, 1e-5))))
            t0 = time.perf_counter()
            for _ in range(n):
                run()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[name] = {"metric": wl["metric"], "value": round(S * n / dt, 2), "unit": wl["unit"], "ms_per_step": round(dt / n * 1e3, 3),
                         "steps": n, "units_per_step": S, "batch": int(x.shape[0]), "dtype": wl["dtype"], "graph_replay": graphed is not None,
                         "workload": wl["describe"], "roofline": secondary_roofline(name, S * n / dt, int(x.shape[0]))}
            del wl, model, x, graphed, run
            torch.cuda.empty_cache()
        except Exception as e:                                       # noqa: BLE001 -- a secondary workload must not take the headline line down
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


SECONDARY_MULTI = (("resnet_bbb_w4", 1024), ("ensemble16", 16))


def secondary_multi(a, world, rank, q, load_golden, seed, fence):
    """N > 1: BASELINE.json's two multi-GPU configs in the SAME driver command as the headline line, after its timed region, on every
    rank -- configs[4] (A7/W4, 1024 MC samples sharded `shard_samples(1024, r, N)`, one RCCL all-reduce of the moments per step) and
    configs[3] (16 SGHMC members over the N ranks).  Strong scaling both (fixed global work); value = global units / MAX-over-ranks time,
    bracketed by barrier + synchronize like the headline."""
    from quantised_bayesian_nets_amd.mc import shard_samples
    out = {}
    for name, units in SECONDARY_MULTI:
        err = None
        dt = 0.0
        n = 10
        try:
            b = argparse.Namespace(**vars(a))
            b.workload, b.samples, b.batch, b.w_bits = name, 0, 256, 8
            wl = WORKLOADS[name](b, world, q, load_golden)
            model, x = wl["model"], wl["x_host"].cuda()
            for _ in range(4):
                wl["step"](model, x, units, seed)
            fence()
            t0 = time.perf_counter()
            for _ in range(n):
                wl["step"](model, x, units, seed)
            fence()
            dt = time.perf_counter() - t0
        except Exception as e:                                       # noqa: BLE001 -- reported below; the collectives that follow keep every rank in step
            err = "%s: %s" % (type(e).__name__, e)
        t = torch.tensor([dt, 1.0 if err else 0.0], dtype=torch.float64, device="cuda")
        tall = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(tall, t)
        if any(float(v[1].item()) for v in tall):
            out[name] = {"error": err or "failed on another rank"}
            continue
        tmax = max(float(v[0].item()) for v in tall)
        out[name] = {"metric": wl["metric"], "value": round(units * n / tmax, 2), "unit": wl["unit"], "ms_per_step": round(tmax / n * 1e3, 3), "steps": n,
                     "units_per_step_global": units, "units_this_rank": shard_samples(units, rank, world)[1], "n_gpus": world, "scaling": "strong",
                     "batch": int(x.shape[0]), "dtype": wl["dtype"], "ms_per_step_by_rank": [round(float(v[0].item()) / n * 1e3, 4) for v in tall],
                     "shards": [list(shard_samples(units, r, world)) for r in range(world)], "workload": wl["describe"]}
        del wl, model, x
        torch.cuda.empty_cache()
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: N fresh ranks under torch.distributed.run as a CHILD process (this
    process has made no HIP call and makes none: never exec from, or fork, a process that initialised the GPU)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def rccl_probe_child(batch, dev_index):
    """Child side of the one-rank RCCL probe (`bench.py --rccl-probe-child B DEV`): prints the mean all-reduce time in microseconds."""
    import socket
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(dev_index)
    last = None
    for _ in range(3):                      # the free-port pick can lose a race: try again with another port
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        try:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0, device_id=torch.device("cuda", dev_index))
            break
        except Exception as e:              # noqa: BLE001
            last = e
    else:
        raise last
    mom = torch.zeros((2, batch, 10), dtype=torch.float64, device="cuda")
    for _ in range(5):
        dist.all_reduce(mom)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        dist.all_reduce(mom)
    torch.cuda.synchronize()
    print("RCCL_ONE_RANK_US %.1f" % ((time.perf_counter() - t) / 50 * 1e6), flush=True)
    dist.destroy_process_group()


def rccl_probe_in_child(batch, dev_index, timeout_s=90):
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--rccl-probe-child", str(batch), str(dev_index)],
                           capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return "unavailable: timeout after %d s" % timeout_s
    for line in r.stdout.splitlines():
        if line.startswith("RCCL_ONE_RANK_US "):
            return float(line.split()[1])
    return "unavailable: child exit %d" % r.returncode


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--rccl-probe-child":
        return rccl_probe_child(int(sys.argv[2]), int(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples", type=int, default=0, help="MC samples per GPU per step (0 = the workload's own: 100 for resnet_bbb, BASELINE configs[2])")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--w-bits", type=int, default=8)
    ap.add_argument("--workload", default="resnet_bbb", choices=sorted(WORKLOADS),
                    help="resnet_bbb = BASELINE.json's metric (default); the others are BASELINE.json's remaining configs")
    ap.add_argument("--prime", type=int, default=12, help="setup steps before the W warm-up steps (clock ramp, allocator)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=0, help="oracle samples to time (0 = auto, about 10-30 s)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline's torch leg (0 = the cores this process may use: affinity, cgroup quota, physical cores)")
    ap.add_argument("--no-graph", action="store_true", help="mlp_f32 (the launch-bound workload: ~25 launches of microseconds): launch eagerly instead of replaying its captured HIP graph")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` block (default workload only; N = 1: BASELINE.json's other configs on one GPU; N > 1: configs[4] = A7/W4 with 1024 global samples and configs[3] = 16 ensemble members, sharded over the N ranks)")
    ap.add_argument("--no-rccl-probe", action="store_true", help="skip the one-rank RCCL all-reduce latency probe of the N = 1 run")
    ap.add_argument("--plumbing-check", action="store_true",
                    help="launcher test (no GPU, gloo): ranks rendezvous, all-reduce their rank, rank 0 prints one JSON line")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if a.plumbing_check:
        # no GPU: the launcher, the rendezvous, the partition of BASELINE configs[4] / configs[3] over the ranks and the one collective
        # of the path (sum all-reduce of the [2, B, C] fp64 moments) over gloo
        from quantised_bayesian_nets_amd.mc import shard_samples, all_reduce_moments
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        mine = {"rank": rank, "samples_1024": list(shard_samples(1024, rank, world)), "members_16": list(shard_samples(16, rank, world))}
        shards = [None] * world
        dist.all_gather_object(shards, mine)
        mom = torch.full((2, a.batch, 10), float(rank + 1), dtype=torch.float64)
        all_reduce_moments(mom)
        dist.barrier()
        if rank == 0:
            # (no RCCL call happens here: the line names the backend it ran on; `rccl_ranks` appears only on lines that used nccl)
            print(json.dumps({"plumbing_check": True, "ranks": dist.get_world_size(), "backend": dist.get_backend(), "rank_sum": float(t.item()),
                              "moments_sum": float(mom[0, 0, 0].item()), "shards": shards,
                              # what `secondary` will hold on a GPU run at this N (secondary_multi): BASELINE configs[4] and configs[3]
                              "secondary_multi": [{"workload": nm, "units_global": u, "shards": [list(shard_samples(u, r, world)) for r in range(world)]}
                                                  for nm, u in SECONDARY_MULTI]}))
        dist.destroy_process_group()
        return
    # rehearsal on a one-GPU box: QBNN_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and QBNN_BENCH_BACKEND=gloo replaces RCCL (which
    # refuses two ranks on one device) -- the N > 1 code path (sharding, barriers, reduce, rank-0 JSON) on the real kernels
    share_gpu = os.environ.get("QBNN_BENCH_SHARE_GPU", "0") == "1"
    backend = os.environ.get("QBNN_BENCH_BACKEND", "nccl")
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    # QBNN_BENCH_FORCE_DIST=1 (under torch.distributed.run with one rank): take the RCCL init / barrier / all-reduce path on a 1-GPU box
    use_dist = world > 1 or os.environ.get("QBNN_BENCH_FORCE_DIST", "0") == "1"
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    import quantised_bayesian_nets_amd as q
    from quantised_bayesian_nets_amd import layers as qlayers
    from fixtures import load_golden           # tests/golden/fixtures.py: plain readers of the committed .npz data

    wl = WORKLOADS[a.workload](a, world, q, load_golden)
    g, model, x_host, step_fn = wl["golden"], wl["model"], wl["x_host"], wl["step"]
    x = x_host.cuda()
    S_local, S_global, seed = wl["units_per_gpu"], wl["units_global"], 3

    graphed = None
    if "graph" in wl and not a.no_graph:
        graphed = q.GraphedPredictor(model, S_global, **wl["graph"])

    def step():
        return graphed(x, seed) if graphed is not None else step_fn(model, x, S_global, seed)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # setup: first launches load the code objects and set kernel attributes, the caching allocator grows to its working set
    # and the GPU clock ramps over the first ~10 steps (5.5 -> 4.2 ms measured); done once here, outside W and K
    for _ in range(a.prime):
        step()
    fence()
    for _ in range(a.warmup):
        step()
    fence()
    # HIP events around the fused conv-block launches only (the candidates for `roofline`; the small kernels are in the
    # rocprofv3 summary under profiles/): every event costs a few microseconds of drained queue inside the timed region
    qlayers.PROFILE_FILTER = lambda meta: bool(meta) and "convs" in meta
    prof = []
    if wl["resnet"]:
        qlayers.PROFILE = prof        # (the launch-bound workloads run as captured graphs: no per-launch events there)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    qlayers.PROFILE = None
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    rank_ms = None
    if use_dist:
        # every rank's own time for the K steps beside the MAX the metric uses: a straggler shows as a spread, not only as a slower job
        tall = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(tall, tmax)
        rank_ms = [round(float(t.item()) / a.steps * 1e3, 4) for t in tall]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- per-kernel times from the HIP events recorded around each launch in the timed region
    agg = {}
    for key, meta, e0, e1 in prof:
        d = agg.setdefault(key, dict(ms=0.0, n=0, meta=meta))
        d["ms"] += e0.elapsed_time(e1)
        d["n"] += 1
    total_ms = sum(d["ms"] for d in agg.values())
    roof = None
    conv_keys = [k for k in agg if agg[k]["meta"] and "convs" in agg[k]["meta"]]
    if conv_keys:
        dom = max(conv_keys, key=lambda k: agg[k]["ms"])
        d = agg[dom]
        m = d["meta"]
        avg_s = d["ms"] / d["n"] * 1e-3
        Bx = x_host.shape[0]
        abytes = sum(conv_algorithmic_bytes(S_local, Bx, H, ci, co, ks, st, nw) for (H, ci, co, ks, st, nw) in m["convs"])
        ops = sum(conv_ops(S_local, Bx, H, ci, co, ks, st) for (H, ci, co, ks, st, nw) in m["convs"])
        gbs, tops = abytes / avg_s / 1e9, ops / avg_s / 1e12
        # HBM bytes per launch: PMC counters need rocprofv3, so they come from the committed passes of tools/profile_round.sh
        # (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc runs) -- but ONLY while that file was measured on this very kernel source
        # and workload shape; otherwise null (never a stale number)
        traffic, traffic_note = None, "no PMC summary for this kernel source: run tools/profile_round.sh"
        # the PMC summary measured on THIS kernel source if there is one, else the latest round's (by round number, not by name order)
        cands = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.match(r"r\d+_pmc_traffic\.json$", f)),
                       key=lambda f: int(re.match(r"r(\d+)_", f).group(1)))
        for f in cands:
            if json.load(open(os.path.join(ROOT, "profiles", f))).get("kernel_source_sha16") == kernel_source_sha16():
                cands = [f]
        tname = cands[-1] if cands else None
        if tname:
            tpath = os.path.join(ROOT, "profiles", tname)
            tj = json.load(open(tpath))
            if tj.get("kernel_source_sha16") != kernel_source_sha16():
                traffic_note = "profiles/%s was measured on an older kernel source" % tname
            elif (tj.get("samples"), tj.get("batch"), tj.get("workload")) != (S_local, x_host.shape[0], a.workload):
                traffic_note = "profiles/%s was measured on another workload shape" % tname
            else:
                traffic, traffic_note = tj["by_bench_key"].get(dom), "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this source)" % tname
        # vector-ALU roof of the exact-arithmetic epilogue: instructions per output element counted in the shipped ISA (6 for a plain
        # requantisation: sub, cvt, fma, mul, min, cvt_pk; 13 for stem.3 + quantized::add + ReLU) x the measured issue cost of that
        # mix at the kernel's occupancy (the 16-wave layer-1 kernel: 4 waves per SIMD; the 8-wave block kernels: 2)
        outs = [S_local * Bx * (H // st) ** 2 * co for (H, ci, co, ks, st, nw) in m["convs"]]
        n_res = sum(1 for i, c in enumerate(m["convs"]) if m.get("res_convs") and i in m["res_convs"])
        w16 = dom.startswith("stem") and os.environ.get("QBNN_W16", "1") != "0"
        magic = w16 and os.environ.get("QBNN_W16_MAGIC", "1") != "0"
        ops_plain, ops_res = (VALU_OPS_PLAIN_MAGIC, VALU_OPS_RES_MAGIC) if magic else (VALU_OPS_PLAIN, VALU_OPS_RES)
        lane_ops = sum(o * (ops_res if (m.get("res_convs") and i in m["res_convs"]) else ops_plain) for i, o in enumerate(outs))
        wps = 4 if w16 else 2
        floor_s = lane_ops / 64.0 * VALU_ISSUE_CYCLES[wps] / 1024 / SHADER_CLOCK_HZ
        valu = {"epilogue_wave_instructions_per_launch": lane_ops / 64.0, "waves_per_simd": wps, "issue_cycles_per_instruction_measured": VALU_ISSUE_CYCLES[wps],
                "floor_ms": round(floor_s * 1e3, 4), "frac": round(floor_s / avg_s, 4),
                "ops_per_output": {"requant": ops_plain, "requant+add+relu": ops_res}, "residual_convs": n_res,
                "note": "issue-slot time of the epilogue's vector instructions alone (profiles/r03_issue_bench.txt); the MFMAs of the same "
                        "launch take further issue slots (~16 cycles each; round 5: 7 per output row in the layer-1 kernel, whose weights are packed with the kernel rows' tails gathered)"}
        common = {"kernel": dom, "avg_launch_ms": round(avg_s * 1e3, 4), "launches": d["n"],
                  "share_of_step_time": round(d["ms"] / (dt * 1e3), 3), "convs_in_launch": len(m["convs"]),
                  "algorithmic_bytes_per_launch": abytes, "algorithmic_ops_per_launch": ops, "traffic": traffic, "traffic_source": traffic_note,
                  "valu": valu}
        if m["fused"]:
            # fused block kernels keep activations in LDS: their HBM traffic is a fraction of the layer-granular byte
            # model, the binding roof is the int8 matrix pipe
            roof = {"bound": "mfma", "achieved": round(tops, 1), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                    "frac": round(tops / MFMA_I8_PEAK_TOPS, 4), "sustained_peak_measured": MFMA_I8_SUSTAINED_TOPS,
                    "frac_of_sustained": round(tops / MFMA_I8_SUSTAINED_TOPS, 4), "layer_granular_GBps": round(gbs, 1),
                    "layer_granular_frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), **common}
        else:
            roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "mfma_achieved_tops": round(tops, 1), "mfma_frac_of_i8_peak": round(tops / MFMA_I8_PEAK_TOPS, 4), **common}
    kernels = {k: {"ms_per_step": round(v["ms"] / a.steps, 3), "launches_per_step": v["n"] // a.steps} for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}

    # ---- whole-path roofline view (layer-granular byte model of SURVEY 8(d): 126.66 MB / sample at B=256, int8)
    value = S_global * a.steps / dt
    path = None
    if wl["resnet"]:
        bytes_per_sample = 126.66e6 * a.batch / 256
        path = {"algorithmic_GBps_per_gpu": round(value / world * bytes_per_sample / 1e9, 1),
                "frac_of_hbm_peak": round(value / world * bytes_per_sample / 1e9 / HBM_PEAK_GBS, 4),
                "int8_TOPS_per_gpu": round(value / world * 40.203e9 * a.batch / 256 / 1e12, 1)}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and wl["cpu_baseline"]:      # the CPU leg runs at N = 1 only
        cpu = cpu_baseline(a, g, x_host, seed)
        # the oracle's sample 0 doubles as an in-run parity check of the GPU result
        with q.mc_context(1, seed, 0):
            p_gpu = model.forward_mc(x)[0].cpu().numpy()         # (resnet workloads only)
        cpu["gpu_matches_oracle_sample0"] = bool(np.allclose(p_gpu, cpu.pop("_p_oracle_sample0"), rtol=1e-5, atol=1e-8))

    secondary = None
    if rank == 0 and world == 1 and a.workload == "resnet_bbb" and not a.no_secondary:
        secondary = secondary_workloads(a, q, load_golden, seed)

    if use_dist and world > 1 and a.workload == "resnet_bbb" and not a.no_secondary:
        secondary = secondary_multi(a, world, rank, q, load_golden, seed, fence)

    rccl = None
    if use_dist:
        # the path's one collective in isolation: sum all-reduce of the [2, B, C] fp64 moments (40 KB), RCCL over xGMI
        mom = torch.zeros((2, a.batch, 10), dtype=torch.float64, device="cuda")
        for _ in range(5):
            dist.all_reduce(mom)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(mom)
        torch.cuda.synchronize()
        rccl = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "allreduce_bytes": mom.numel() * 8,
                "allreduce_us": round((time.perf_counter() - t) / 50 * 1e6, 1), "collectives_per_step": 1}

    rccl_one_rank_us = None
    if not use_dist and world == 1 and not a.no_rccl_probe:
        # Multi-GPU readiness on a one-GPU box: the path's one collective (sum all-reduce of the [2, B, C] fp64 moments, 40 KB) through
        # RCCL with a single rank -- a reference point for the first 8-GPU run (there is no curve to report here).  It runs in a FRESH
        # CHILD process under a timeout: an RCCL bootstrap that hangs or aborts (no exception to catch) must not cost the bench line.
        rccl_one_rank_us = rccl_probe_in_child(a.batch, dev_index)

    if rank == 0:
        out = {"metric": wl["metric"], "value": round(value, 2), "unit": wl["unit"],
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": wl["scaling"], "vs_baseline": None, "dtype": wl["dtype"], "data": "synthetic",
               "config": {"workload": wl["describe"], "samples_per_gpu": S_local, "global_samples": S_global, "batch": x_host.shape[0],
                          "image_samples_per_s": round(value * x_host.shape[0], 1), "parallelism": f"mc-sample-shard x{world}"},
               "graph_replay": graphed is not None, "roofline": roof, "path_roofline": path, "cpu_baseline": cpu, "rccl": rccl, "rccl_ranks": world if (use_dist and backend == "nccl") else 0, "rccl_one_rank_us": rccl_one_rank_us,
               "ms_per_step_by_rank": rank_ms, "ms_per_step_min_max_over_ranks": ([min(rank_ms), max(rank_ms)] if rank_ms else None),
               "kernels": kernels, "secondary": secondary}
        print(json.dumps(out), flush=True)          # the line is out before any process-group teardown
    if use_dist:
        dist.destroy_process_group()
    from quantised_bayesian_nets_amd import models as qmodels
    if qmodels.GRAPH_FALLBACKS:        # a refused HIP-graph capture silently changes what was timed: fail the run instead
        print("bench.py: HIP graph capture fell back to eager launches: %s" % "; ".join(qmodels.GRAPH_FALLBACKS), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
