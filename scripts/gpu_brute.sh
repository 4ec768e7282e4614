#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
for CUT in 1 0; do
SPM_HIP_BRUTE_CUTOFF=$CUT timeout -k 10 300 python bench.py --engine brute --text-gib 0.5 --steps 3 --warmup 1 --no-cpu-baseline --packed-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('myers CUT=$CUT brute Gbases/s', round(r['value'],3), 'ms', round(r['ms_per_step'],2), r['hits'], r['all_planted_found'])
"
done
timeout -k 10 300 python bench.py --workload c2 --engine brute --text-gib 0.5 --steps 3 --warmup 1 --no-cpu-baseline --packed-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('shiftor brute Gbases/s', round(r['value'],3), 'ms', round(r['ms_per_step'],2), r['hits'], r['all_planted_found'])
"
