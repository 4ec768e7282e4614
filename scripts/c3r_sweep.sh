#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for f in ${FRACS:-0.01 0.05}; do for e in ${EVERY:-64 8}; do python bench.py --workload c3r --repeat-frac $f --repeat-needle-every $e --steps 5 --warmup 3 --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('c3r f=$f e=$e RUNS=${SPM_HIP_VERIFY_RUNS:-1}', round(r['value'],1), 'Gbases/s', round(r['ms_per_step'],3), 'ms kernel', round(r['roofline']['kernel_ms'],3), 'verify', round(r['verify_ms_per_step'],3), 'cand', r['candidates'], 'bands', r['bands_verified'], 'hits', r['hits'], r['parity_slice']['equal_to_brute_force_engine'])"; done; done
