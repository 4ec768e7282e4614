#!/bin/bash
# resolve_kernel stage by stage on the repeat text: duration of the kernel cut short after stage N (SPM_HIP_RESOLVE_DEBUG;
# 1 survivors + text window + directory, 2 + dealing and seed signatures, 3 + whole-seed check, 4 + piece counts, 0 all)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 --no-other-configs --workload c3r --steps 3 --warmup 2"
: > gpurun_out/resolve_stages.log
for f in 0.01 0.05; do for st in 1 2 3 4 0; do
  rm -rf gpurun_out/rs_trace
  (cd /tmp && SPM_HIP_RESOLVE_DEBUG=$st rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rs_trace -- $B --repeat-frac $f > /dev/null 2>&1)
  python3 - <<PY | tee -a gpurun_out/resolve_stages.log
import csv, glob
f = sorted(glob.glob("gpurun_out/rs_trace/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "resolve_kernel" in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows)
full = [x for x in d if x > 0.5 * d[-1]]
print("frac $f stage $st: resolve_kernel", round(sorted(full)[len(full) // 2], 3), "ms (median of", len(full), "full launches)")
PY
done; done
