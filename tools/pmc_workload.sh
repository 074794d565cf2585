#!/bin/bash
# usage (GPU box, repo root): tools/pmc_workload.sh <round tag> <workload> ...  -- per-kernel MFMA-busy / VALU-issue / wave-state shares of a secondary
# workload (rows a1 / a2: the "stated roofline" of the fp32 / fp64 convs) -> gpurun_out/wl/<tag>_pmc_util_<workload>.json
R=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wl
for W in "$@"; do
  rm -rf gpurun_out/wl/pmc_$W
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/wl/pmc_$W/a -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-probe > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d gpurun_out/wl/pmc_$W/b -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-probe > /dev/null 2>&1
  QBNN_W=$W QBNN_ROUND=$R python3 - <<'PY'
import csv, glob, collections, json, os
W, R = os.environ["QBNN_W"], os.environ["QBNN_ROUND"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/wl/pmc_%s/*/*/*counter_collection.csv" % W):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    v = {n: sum(x) / len(x) for n, x in c.items()}
    v["launches_seen"] = max(len(x) for x in c.values())
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8          # the counter sums the 8 XCDs
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if cyc:
        v["shader_cycles"] = cyc
        v["valu_issue_frac(4cyc/instr)"] = v.get("SQ_INSTS_VALU", 0) * 4 / (cyc * 1024)
        v["mfma_busy_frac"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024)
        v["lds_active_frac"] = v.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * 256)
    if wc:
        v["wave_parked_frac"] = v.get("SQ_WAIT_ANY", 0) / wc
        v["wave_issue_stalled_frac"] = v.get("SQ_WAIT_INST_ANY", 0) / wc
        v["wave_issuing_frac"] = v.get("SQ_ACTIVE_INST_ANY", 0) / wc
    out[k[:110]] = v
json.dump(out, open("gpurun_out/wl/%s_pmc_util_%s.json" % (R, W), "w"), indent=1)
print(W)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("shader_cycles", 0) * kv[1].get("launches_seen", 1))[:6]:
    print("  %-70s x%-4d cyc %9.0f  MFMA busy %5.1f%%  VALU issue %5.1f%%  parked %.2f stalled %.2f issuing %.2f" % (k[:70], v["launches_seen"], v.get("shader_cycles", 0),
          100 * v.get("mfma_busy_frac", 0), 100 * v.get("valu_issue_frac(4cyc/instr)", 0), v.get("wave_parked_frac", 0), v.get("wave_issue_stalled_frac", 0), v.get("wave_issuing_frac", 0)))
PY
done
