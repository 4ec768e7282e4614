#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
timeout -k 10 120 ./tests/cpp/reference_cases 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
timeout -k 10 600 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 2>&1 | tee $OUT/bench_c4.json | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('c4', round(r['value'],2), round(r['ms_per_step'],3), round(r['roofline']['kernel_ms'],3), r['hits'], r['needles_found'], r['candidates'], r['verify_ms_per_step'], r['fell_back'])
    else: print(l.rstrip())
"
