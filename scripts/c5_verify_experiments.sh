#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for nb in 1 2; do for kb in 0 20 40; do
SPM_HIP_VERIFY_WAVE_NB=$nb SPM_HIP_VERIFY_WAVE_LDS_KB=$kb python bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline --brute-sample-mib 0 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('c5 NB=$nb LDS_KB=$kb', 'ms/step', round(r['ms_per_step'],4), 'kernel', round(r['roofline']['kernel_ms'],3), 'verify(resolve+select+verify)', round(r['verify_ms_per_step'],3), 'hits', r['hits'])"
done; done
