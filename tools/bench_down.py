"""Times one fused down-sampling block launch in isolation (d24 / d48 / d96; B = 256, S = 100).  A/B runs: build variant libraries with
QBNN_LIB_OVERRIDE=tools/_build/<name>.so and pass them all in ONE gpurun call (boxes differ by ~10 % in clock):
  for l in a b; do QBNN_LIB_OVERRIDE=tools/_build/libqbnn_$l.so python tools/bench_down.py d48 d96; done"""
import sys, types, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
from fixtures import load_golden
import quantised_bayesian_nets_amd as q
from quantised_bayesian_nets_amd.layers import MCQTensor, sample_all_weights
from quantised_bayesian_nets_amd.models import run_down_block, run_identity_chain
g = load_golden('resnet_bbb_a7w8.npz')
args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
m = q.ModelFactory.get_model('conv_resnet_bbb', [1, 3, 32, 32], 10, True, args).load_reference_state(g['state'])
S, B = 100, 256
cases = {'d24': (m.layers[4][0], (S, B, 32, 32, 24)), 'd48': (m.layers[5][0], (S, B, 16, 16, 48)), 'd96': (m.layers[6][0], (S, B, 8, 8, 96)),
         'c48': (m.layers[4][1], (S, B, 16, 16, 48)), 'c96': (m.layers[5][1], (S, B, 8, 8, 96)), 'c192': (m.layers[6][1], (S, B, 4, 4, 192))}
from quantised_bayesian_nets_amd import layers as ql
out = []
for which in sys.argv[1:]:
    blk, xs = cases[which]
    x = MCQTensor(torch.randint(0, 128, xs, dtype=torch.uint8, device='cuda'), 0.05, 60)
    run = (lambda: run_down_block(blk, x)) if which[0] == 'd' else (lambda: run_identity_chain([blk], x))
    with q.mc_context(S, 3, 0):
        for _ in range(5): run()
        torch.cuda.synchronize()
        ql.PROFILE = []
        for _ in range(20): run()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for key, meta, e0, e1 in ql.PROFILE if key.startswith('block_'))
        ql.PROFILE = None
    out.append('%s %.4f ms (median of 20 launches, HIP events)' % (which, ts[len(ts) // 2]))
import os
print(os.path.basename(os.environ.get('QBNN_LIB_OVERRIDE', 'libqbnn_hip.so')), ' | '.join(out))
