"""Per-phase s_memtime sums of the ring form of the wide down-sampling blocks (csrc/qbnn_down_ring.hip); needs a diagnostic library:
QBNN_HIPCC_EXTRA="-DQBNN_STAMP -DQBNN_STAMP_WAVE=0" QBNN_LIB_OVERRIDE=tools/_build/libqbnn_STAMP0.so python -m quantised_bayesian_nets_amd.build"""
import sys, os, types, ctypes as C, numpy as np, torch
os.environ["QBNN_LIB_OVERRIDE"] = os.path.abspath(os.environ.get("QBNN_STAMP_LIB", "tools/_build/libqbnn_STAMP0.so"))
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
from fixtures import load_golden
import quantised_bayesian_nets_amd as q
from quantised_bayesian_nets_amd import _lib
from quantised_bayesian_nets_amd.layers import MCQTensor
from quantised_bayesian_nets_amd.models import run_down_block, run_identity_chain
g = load_golden('resnet_bbb_a7w8.npz')
args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
m = q.ModelFactory.get_model('conv_resnet_bbb', [1, 3, 32, 32], 10, True, args).load_reference_state(g['state'])
S, B = 100, 256
which = sys.argv[1]
cases = {'d48': (m.layers[5][0], (S, B, 16, 16, 48), 4), 'd96': (m.layers[6][0], (S, B, 8, 8, 96), 8),
         'c96': (m.layers[5][1], (S, B, 8, 8, 96), 8), 'c192': (m.layers[6][1], (S, B, 4, 4, 192), 16)}
blk, xs, G = cases[which]
x = MCQTensor(torch.randint(0, 128, xs, dtype=torch.uint8, device='cuda'), 0.05, 60)
dbg = torch.zeros(64, dtype=torch.int64, device='cuda')
L = _lib.lib()
chain = which[0] == 'c'
run = (lambda: run_identity_chain([blk], x)) if chain else (lambda: run_down_block(blk, x))
setbuf = L.qbnn_debug_stamp_buffer_chain_ring if chain else L.qbnn_debug_stamp_buffer_ring
setbuf.argtypes = [C.c_void_p]
with q.mc_context(S, 3, 0):
    run(); torch.cuda.synchronize()
    setbuf(C.c_void_p(dbg.data_ptr()))
    inner = (C.c_ulonglong * 4)(); L.qbnn_debug_read_inner_ring(inner)
    run(); torch.cuda.synchronize()
    L.qbnn_debug_read_inner_ring(inner)
d = dbg.cpu().numpy().reshape(8, 8).astype(np.float64)
n_items = S * B / G
names = ['M_a', 'barrier', 'E_a', 'M_b', 'barrier', 'E_b', 'barrier', 'read-out + next X'] if chain else ['M_a', 'M_s', 'barrier', 'E_a + E_s', 'M_b', 'E_b', 'barrier', 'read-out + barrier + next X']
print(which, 'cycles per item (%d images), waves 0 / 3 / 4 / 7' % G)
for i, n in enumerate(names): print('%-28s %8.0f %8.0f %8.0f %8.0f' % (n, d[0, i] / n_items, d[3, i] / n_items, d[4, i] / n_items, d[7, i] / n_items))
print('total', d[0].sum() / n_items, ' | stamped wave, all M phases per item: slab barrier wait %.0f, slab compute %.0f' % (inner[0] / n_items, inner[1] / n_items))
