"""Timeline of one work item of the 16-wave 48-channel chain kernel (diagnostic library built with -DQBNN_C48_STAMP):
    QBNN_HIPCC_EXTRA=-DQBNN_C48_STAMP QBNN_LIB_OVERRIDE=$PWD/tools/_ab/libqbnn_C48STAMP.so python -m quantised_bayesian_nets_amd.build
    QBNN_STAMP_LIB=tools/_ab/libqbnn_C48STAMP.so python tools/stamp_c48.py        (on the GPU box)
Prints, for the four waves of each SIMD (waves w, w+4, w+8, w+12), the s_memtime ticks since the workgroup's first stamp and the step
since the wave's previous stamp.  s_memtime runs at a constant 100 MHz: x (shader clock / 100 MHz) gives shader cycles."""
import sys, os, types, ctypes as C, numpy as np, torch
os.environ["QBNN_LIB_OVERRIDE"] = os.path.abspath(os.environ.get("QBNN_STAMP_LIB", "tools/_ab/libqbnn_C48STAMP.so"))
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
from fixtures import load_golden
import quantised_bayesian_nets_amd as q
from quantised_bayesian_nets_amd import _lib
g = load_golden('resnet_bbb_a7w8.npz')
args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
m = q.ModelFactory.get_model('conv_resnet_bbb', [1, 3, 32, 32], 10, True, args).load_reference_state(g['state'])
S, B = 100, 256
x = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(2)).cuda()
dbg = torch.zeros(16 * 32, dtype=torch.int64, device='cuda')
L = _lib.lib(); L.qbnn_debug_c48_stamp_buffer.argtypes = [C.c_void_p]
with q.mc_context(S, 3, 0):
    m.forward_mc(x); torch.cuda.synchronize()
    L.qbnn_debug_c48_stamp_buffer(C.c_void_p(dbg.data_ptr()))
    m.forward_mc(x); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(16, 32)
n = int((d[0] != 0).sum())
t0 = d[:, 0].min()
labels = ["top", "barrier 1", "a: M issued", "a: mid", "a: E0", "a: E1", "barrier 2", "b: M issued", "b: mid (X write)", "b: E0", "b: E1", "end"]
for w0 in range(4):
    ws = [w0, w0 + 4, w0 + 8, w0 + 12]
    print("waves", ws)
    for k in range(n):
        row = "%-18s" % (labels[k] if k < len(labels) else "?")
        for w in ws:
            row += " %7d (%5d)" % (d[w, k] - t0, d[w, k] - d[w, k - 1] if k else 0)
        print(row)
print("item total:", (d[:, n - 1] - d[:, 0]).max(), "ticks")
