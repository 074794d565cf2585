"""conv_resnet_mc at B = 256, S = 100: dropout / Add fused into the convs' store passes against one launch per op."""
import os, types, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import quantised_bayesian_nets_amd as q
from quantised_bayesian_nets_amd import models_mc
from fixtures import load_golden
g = load_golden("resnet_mc_a7w8.npz")
args = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=g["meta"]["p"])
m = q.ModelFactory.get_model("conv_resnet_mc", [1, 3, 32, 32], 10, True, args).load_reference_state(g["state"])
x = torch.randn(256, 3, 32, 32).cuda()
for fused in (False, True):
    models_mc.BasicBlock.fuse_post = fused
    for _ in range(3): q.mc_predict(m, x, 100, 1)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): q.mc_predict(m, x, 100, 1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print("fused" if fused else "separate", "%.3f ms per 100 samples, %.1f k MC samples/s" % (dt * 1e3, 0.1 / dt))
