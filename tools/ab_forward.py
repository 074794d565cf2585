"""A/B timing of the int8 BBB ResNet forward (B = 256, S = 100), one process per library variant:
    for l in base x; do QBNN_LIB_OVERRIDE=$PWD/tools/_ab/libqbnn_$l.so python tools/ab_forward.py [key-substring ...]; done
Prints the median HIP-event time of every fused launch whose PROFILE key contains one of the substrings (default: all block launches),
the step time (wall clock over 20 steps, no events) and a hash of the probabilities -- variants must print the same hash.
All variants of a comparison go into ONE gpurun call: boxes differ by 5 - 10 % in clock."""
import hashlib, os, sys, time, types
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
from fixtures import load_golden
import quantised_bayesian_nets_amd as q
from quantised_bayesian_nets_amd import layers as ql

subs = sys.argv[1:] or ["block_", "stem"]
g = load_golden('resnet_bbb_a7w8.npz')
args = types.SimpleNamespace(activation_precision=int(os.environ.get("AB_ABITS", "7")), weight_precision=int(os.environ.get("AB_WBITS", "8")))
m = q.ModelFactory.get_model('conv_resnet_bbb', [1, 3, 32, 32], 10, True, args).load_reference_state(g['state'])
S, B = int(os.environ.get("AB_S", "100")), int(os.environ.get("AB_B", "256"))
x = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(2)).cuda()
with q.mc_context(S, 3, 0):
    for _ in range(8):
        p = m.forward_mc(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        p = m.forward_mc(x)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 20 * 1e3
    ql.PROFILE = []
    for _ in range(15):
        m.forward_mc(x)
    torch.cuda.synchronize()
    ts = {}
    for key, meta, e0, e1 in ql.PROFILE:
        if any(s in key for s in subs):
            ts.setdefault(key, []).append(e0.elapsed_time(e1))
    ql.PROFILE = None
name = os.path.basename(os.environ.get('QBNN_LIB_OVERRIDE', 'libqbnn_hip.so'))
envs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("QBNN_") and k != "QBNN_LIB_OVERRIDE")
h = hashlib.sha256(p.cpu().numpy().tobytes()).hexdigest()[:12]
print("%-28s %s step %.3f ms  probs %s" % (name, envs, step, h))
tot = 0.0
for key, v in ts.items():
    v.sort()
    tot += v[len(v) // 2]
    print("    %-44s %.4f ms" % (key, v[len(v) // 2]))
print("    %-44s %.4f ms" % ("sum of the listed launches", tot))
