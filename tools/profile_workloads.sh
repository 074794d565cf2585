#!/bin/bash
# usage (GPU box, repo root): tools/profile_workloads.sh <round tag> <workload> [...]  -- per workload: the bench line of `bench.py --workload W` and the
# rocprofv3 --kernel-trace --stats summary of the same command, into gpurun_out/wl/ (copy what is judged to profiles/).
R=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wl
for W in "$@"; do
  timeout -k 10 200 python3 bench.py --workload $W --steps 10 --warmup 6 --no-cpu-baseline --no-secondary --no-rccl-probe > gpurun_out/wl/${R}_workload_$W.json 2> gpurun_out/wl/${R}_workload_$W.err || { echo "bench $W failed"; tail -3 gpurun_out/wl/${R}_workload_$W.err; }
  rm -rf gpurun_out/wl/stats_$W
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wl/stats_$W -- python3 bench.py --workload $W --steps 10 --warmup 6 --no-cpu-baseline --no-secondary --no-rccl-probe > /dev/null 2>&1
  for f in gpurun_out/wl/stats_$W/*/*kernel_stats.csv; do cp $f gpurun_out/wl/${R}_kernel_stats_$W.csv; done
  python3 - <<PY
import json, csv
d = json.loads(open("gpurun_out/wl/${R}_workload_$W.json").read().strip().splitlines()[-1])
print("$W", d["value"], d["unit"], d["ms_per_step"], "ms/step")
rows = list(csv.DictReader(open("gpurun_out/wl/${R}_kernel_stats_$W.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print("   %5.1f%% %7d calls %9.1f us avg  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
  rm -rf gpurun_out/wl/stats_$W
done
