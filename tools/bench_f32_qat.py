"""Float BBB ResNet-18 (row a1) and QAT evaluation with live observers (row a2) at B = 256: samples/s; python tools/bench_f32_qat.py [f32|qat] [S]"""
import os, sys, time, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import quantised_bayesian_nets_amd as q
from fixtures import load_golden
which = sys.argv[1] if len(sys.argv) > 1 else "qat"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10
x = torch.randn(256, 3, 32, 32, generator=torch.Generator().manual_seed(2)).cuda()
if which == "qat":
    g = load_golden("resnet_bbb_qat.npz")
    args = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=8, qat_eval=True)
    m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args).load_reference_state(g["state"])
else:
    g = load_golden("resnet_bbb_f32.npz")
    args = types.SimpleNamespace(sigma_prior=-2.0)
    m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, False, args).load_reference_state(g["state"])
def run():
    with q.mc_context(S, 1, 0):
        return m.forward_mc(x)
for _ in range(3): run()
torch.cuda.synchronize(); t = time.perf_counter()
n = 5
for _ in range(n): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print(which, "S=%d: %.2f ms per pass, %.0f samples/s" % (S, dt * 1e3, S / dt))
