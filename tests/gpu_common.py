"""Helpers shared by the GPU parity test modules (tests/test_gpu_*.py; split by SURVEY section-8 row in round 6)."""
import ctypes as C
import os
import types

import numpy as np
import pytest
import torch

RTOL = 1e-5


def _args(g):
    return types.SimpleNamespace(activation_precision=g["meta"]["a_bits"], weight_precision=g["meta"]["w_bits"])


def _model(g):
    import quantised_bayesian_nets_amd as q
    m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, _args(g))
    return m.load_reference_state(g["state"])


def _pack_per_sample(L, w, layout=0):
    """int8 [S, Cout, KH, KW, Cin] -> QBNN_LAYOUT_MFMA32 (0) / _MFMA32_N24 (2) fragments [S, nbytes] on the device (krow as layers.Conv2d chooses it)."""
    import ctypes as C
    from quantised_bayesian_nets_amd import _lib
    S, cout, kh, kw, cin = w.shape
    k = kh * kw * cin
    krow = kw * cin if cin % 8 == 0 else k
    nb = L.qbnn_packed_weight_bytes(cout, k, krow, layout)
    out = np.zeros((S, nb), np.int8)
    for s in range(S):
        src = np.ascontiguousarray(w[s].reshape(cout, k))
        _lib.check(L.qbnn_pack_weights_host(src.ctypes.data_as(C.c_void_p), cout, k, krow, layout, out[s].ctypes.data_as(C.c_void_p)))
    return torch.from_numpy(out).cuda(), nb
