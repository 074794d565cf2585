"""Native prepare -> calibrate -> convert against the reference's recorded run, observer by observer, next to the reference's OWN
run-to-run spread (tests/golden/resnet_bbb_prepare_spread.npz: mkldnn on / off, 1 / 3 / 8 threads).  GPU box."""
import os, sys, types, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import quantised_bayesian_nets_amd as q
from quantised_bayesian_nets_amd.convert import prepare_model_state, calibrate, convert_model_state
G = os.path.join(ROOT, "tests", "golden")
d = np.load(os.path.join(G, "resnet_bbb_f32.npz"))
fstate = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
ref = np.load(os.path.join(G, "resnet_bbb_prepare_calibrate.npz"))
spr = np.load(os.path.join(G, "resnet_bbb_prepare_spread.npz"))
S, seed = int(ref["meta.samples"]), int(ref["meta.philox_seed"])
aq = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=8, qat_eval=True)
m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, aq).load_reference_state(prepare_model_state(fstate))
x = torch.from_numpy(d["x"]).cuda()
calibrate(m, [x] * S, seed)
st = m.prepared_state()
rows = []
for k in ref.files:
    if not k.startswith("calibrated/") or not k.endswith("min_val"): continue
    kk = k[len("calibrated/"):]
    if any(t in kk for t in ("weight_fake_quant", "std_fake_quant", "mul_noise", "add_weight")): continue
    lo, hi = float(ref[k]), float(ref[k.replace("min_val", "max_val")])
    glo, ghi = float(st[kk]), float(st[kk.replace("min_val", "max_val")])
    rng = max(hi, 0.0) - min(lo, 0.0)
    dev = max(abs(glo - lo), abs(ghi - hi)) / rng
    rows.append((kk[:-len("min_val")], dev, float(spr["spread/" + kk[:-len("min_val")] + "range_frac"])))
for name, dev, sp in rows:
    print("%-60s ours %.4f   reference spread %.4f" % (name, dev, sp))
print("max ours %.4f, max reference spread %.4f" % (max(r[1] for r in rows), max(r[2] for r in rows)))

cfg = [str(c) for c in spr["meta.configs"]]
conv = convert_model_state(st, types.SimpleNamespace(activation_precision=7, weight_precision=8))
for i, c in enumerate(cfg):
    worst, wk = 0.0, ""
    for name, _, _ in rows:
        lo, hi = float(spr["run%d/calibrated/%smin_val" % (i, name)]), float(spr["run%d/calibrated/%smax_val" % (i, name)])
        glo, ghi = float(st[name + "min_val"]), float(st[name + "max_val"])
        rng = max(hi, 0.0) - min(lo, 0.0)
        dev = max(abs(glo - lo), abs(ghi - hi)) / rng
        if dev > worst: worst, wk = dev, name
    sc = zp = 0.0
    for k in spr.files:
        if k.startswith("run%d/converted/" % i):
            key = k[len("run%d/converted/" % i):]
            if key not in conv: continue
            v = float(np.asarray(conv[key]).reshape(-1)[0]); r = float(spr[k])
            if key.endswith("scale"): sc = max(sc, abs(v - r) / abs(r))
            else: zp = max(zp, abs(v - r))
    print("%-22s max observer deviation %.3e of range (%s); converted scales max rel %.3e, zero points max abs %g" % (c, worst, wk, sc, zp))
