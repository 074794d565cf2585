#!/bin/bash
# Round-end evidence: bench line, rocprofv3 kernel stats of the same command, HBM-traffic and utilisation counters.
# usage (GPU box, repo root): tools/profile_round.sh [round tag, default r03]
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
rm -rf gpurun_out/final/stats gpurun_out/final/pmc_*
timeout 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-rccl-probe > gpurun_out/final/bench_prof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final/pmc_f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-probe > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/final/pmc_w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-probe > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/final/pmc_a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-probe > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/final/pmc_b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-probe > /dev/null 2>&1
QBNN_ROUND=$R python3 - <<'PY'
import csv, glob, collections, json, shutil, hashlib, os
R = os.environ.get('QBNN_ROUND', 'r04')
def src_sha():
    import sys
    sys.path.insert(0, os.getcwd())
    from quantised_bayesian_nets_amd import build as _b
    return _b.kernel_source_sha16()
def bench_key(k):
    rules = [("stem_chain_w16_kernel", "stem + block_chain_i8 x2 32x32 c24"), ("block_chain_ws_kernel<ConvCfg<24, 24", "stem + block_chain_i8 x2 32x32 c24"), ("block_chain_pp_kernel<ConvCfg<48", "block_chain_i8 x1 16x16 c48"), ("chain48_w16_kernel", "block_chain_i8 x1 16x16 c48"), ("down24_w16_kernel", "block_down_i8 32x32 24->48"),
             ("block_chain_ws_kernel<ConvCfg<48", "block_chain_i8 x1 16x16 c48"), ("block_chain_ald_kernel<ConvCfg<96", "block_chain_i8 x1 8x8 c96"),
             ("block_chain_ald_kernel<ConvCfg<192", "block_chain_i8 x1 4x4 c192"), ("block_down_ws_kernel<ConvCfg<24", "block_down_i8 32x32 24->48"),
             ("block_down_ws_kernel<ConvCfg<48", "block_down_i8 16x16 48->96"), ("block_down_ws_kernel<ConvCfg<96", "block_down_i8 8x8 96->192"),
             ("block_chain_ring_kernel<ConvCfg<96", "block_chain_i8 x1 8x8 c96"), ("block_chain_ring_kernel<ConvCfg<192", "block_chain_i8 x1 4x4 c192"),
             ("block_down_ring_kernel<(anonymous namespace)::DRCfg<48", "block_down_i8 16x16 48->96"), ("block_down_ring16_kernel<(anonymous namespace)::DRCfg<48", "block_down_i8 16x16 48->96"), ("block_down_ring16_kernel<(anonymous namespace)::DRCfg<96", "block_down_i8 8x8 96->192"), ("block_down_ring_kernel<(anonymous namespace)::DRCfg<96", "block_down_i8 8x8 96->192"),
             ("sample_weights_multi_kernel", "sample_weights_i8_multi"), ("head_i8_kernel", "head_i8"), ("reduce_moments_kernel", "reduce_moments"),
             ("im2col3x3_c3_kernel", "im2col3x3_c3"), ("quantize_input_kernel", "quantize_input"), ("conv_i8_kernel", "conv_i8 32x32 3->24 k3 s1")]
    for pat, key in rules:
        if pat in k:
            return key
    return None
raw = {}
for tag, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    for f in glob.glob("gpurun_out/final/pmc_%s/*/*counter_collection.csv" % tag):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            raw.setdefault(k, {})[name + "_KiB_per_launch"] = sum(v) / len(v)
            raw[k]["launches"] = len(v)
by_key = {}
for k, v in raw.items():
    # gfx950: FETCH_SIZE reports half the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section) -> x2
    v["hbm_bytes_per_launch_corrected"] = (2 * v.get("FETCH_SIZE_KiB_per_launch", 0) + v.get("WRITE_SIZE_KiB_per_launch", 0)) * 1024
    bk = bench_key(k)
    if bk:
        by_key[bk] = v["hbm_bytes_per_launch_corrected"]
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1` (S=100, B=256); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section)",
           "kernel_source_sha16": src_sha(), "workload": "resnet_bbb", "samples": 100, "batch": 256,
           "by_bench_key": by_key, "raw": raw}, open("gpurun_out/final/%s_pmc_traffic.json" % R, "w"), indent=1)
print("HBM bytes per step:", sum(by_key.values()) / 1e9, "GB")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for part in "ab":
    for f in glob.glob("gpurun_out/final/pmc_%s/*/*counter_collection.csv" % part):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
util = {}
for k, c in acc.items():
    v = {n: sum(x) / len(x) for n, x in c.items()}
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc > 0:
        v["shader_cycles"] = cyc
        v["valu_issue_frac(4cyc/instr)"] = v.get("SQ_INSTS_VALU", 0) * 4 / (cyc * 1024)
        v["mfma_busy_frac"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024)
        v["lds_active_frac"] = v.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * 256)
        v["lds_conflict_share"] = v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1)
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if wc > 0:       # of a resident wave's cycles (quad-cycles; the three are disjoint and add up to ~1): parked at s_waitcnt / s_barrier, stalled at issue, issuing
        v["wave_parked_frac"] = v.get("SQ_WAIT_ANY", 0) / wc
        v["wave_issue_stalled_frac"] = v.get("SQ_WAIT_INST_ANY", 0) / wc
        v["wave_issuing_frac"] = v.get("SQ_ACTIVE_INST_ANY", 0) / wc
        v["wait_inst_lds_frac"] = v.get("SQ_WAIT_INST_LDS", 0) / wc
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
        v["valu_mfma_coexec_share_of_mfma_busy"] = v.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / v["SQ_VALU_MFMA_BUSY_CYCLES"]
    util[bench_key(k) or k[:60]] = v
json.dump(util, open("gpurun_out/final/%s_pmc_util.json" % R, "w"), indent=1)
for k, v in sorted(util.items(), key=lambda kv: -kv[1].get("shader_cycles", 0))[:10]:
    print("%-40s cyc %8.0f VALU %5.1f%% MFMA %5.1f%% LDS %5.1f%%" % (k, v.get("shader_cycles", 0), 100 * v.get("valu_issue_frac(4cyc/instr)", 0), 100 * v.get("mfma_busy_frac", 0), 100 * v.get("lds_active_frac", 0)))
for f in glob.glob("gpurun_out/final/stats/*/*kernel_stats.csv"):
    shutil.copy(f, "gpurun_out/final/%s_kernel_stats.csv" % R)
PY
cat gpurun_out/final/bench.json
