#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
for w in c3 c2; do
timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --brute-sample-mib 0 2>&1 | tee $OUT/bench_$w.json | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$w', round(r['value'],1), round(r['ms_per_step'],3), round(r['roofline']['kernel_ms'],3), round(r['roofline']['frac'],4), r['hits'], r.get('packed_text_shadow'))
    else: print(l.rstrip())
"
done
