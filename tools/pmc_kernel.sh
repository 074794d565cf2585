#!/bin/bash
# usage (GPU box, repo root): tools/pmc_kernel.sh <out-tag> [env assignments ...]   -- SQ counters of every kernel of one bench step
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$TAG
for v in "$@"; do export "$v"; done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-probe > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG/b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-probe > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG/c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-probe > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_$TAG/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    v = {n: sum(x) / len(x) for n, x in c.items()}
    out[k[:90]] = v
json.dump(out, open("gpurun_out/pmc_$TAG/summary.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:9]:
    wc = v.get("SQ_WAVE_CYCLES", 1)
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k[:70])
    print("   wave-cycles: wait_any %.2f  wait_inst %.2f  active %.2f | valu-active/wave-cycles %.2f | insts valu %.3g lds %.3g mfma %.3g salu %.3g" % (
        v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_ACTIVE_INST_VALU", 0) / wc,
        v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_LDS", 0), v.get("SQ_INSTS_MFMA", 0), v.get("SQ_INSTS_SALU", 0)))
    print("   coexec/mfma_busy %.2f  vmem level (avg in flight per wave) %.2f  lds level %.2f  active vmem %.3f sca %.3f misc %.3f" % (
        v.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / max(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 1), 1), v.get("SQ_INST_LEVEL_VMEM", 0) / wc, v.get("SQ_INST_LEVEL_LDS", 0) / wc,
        v.get("SQ_ACTIVE_INST_VMEM", 0) / wc, v.get("SQ_ACTIVE_INST_SCA", 0) / wc, v.get("SQ_ACTIVE_INST_MISC", 0) / wc))
    if cyc:
        print("   shader cycles %.0f: valu issue(4c) %.2f mfma busy %.2f lds active %.2f conflict share %.2f wait_inst_lds/wave-cyc %.2f" % (
            cyc, v.get("SQ_INSTS_VALU", 0) * 4 / (cyc * 1024), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024), v.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * 256),
            v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1), v.get("SQ_WAIT_INST_LDS", 0) / wc))
PY
