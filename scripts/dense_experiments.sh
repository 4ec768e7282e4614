#!/bin/bash
# dense-pass stage timing: kernel time of the C4 scan with level 1b dropped (DEBUG=1) and with level 1 dropped too (DEBUG=3)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${W:-c4}
for dbg in ${DBGS:-0 1 3 4}; do
  SPM_HIP_DENSE_DEBUG=$dbg timeout -k 10 300 python bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --packed-steps 0 --brute-sample-mib 0 > gpurun_out/dense_dbg_$dbg.log 2>&1
  python - <<PY
import json
r=json.loads(open("gpurun_out/dense_dbg_$dbg.log").read().strip().splitlines()[-1])
print("$W debug=$dbg", round(r["ms_per_step"],3), "ms kernel", round(r["roofline"]["kernel_ms"],3), "hits", r["hits"])
PY
done
