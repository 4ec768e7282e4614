#!/bin/bash
# quick check: GPU parity suite + C3/C2/C4 one-liners
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
for w in c3 c2 c4; do
  extra=""; [ $w = c4 ] && extra="--steps 3 --warmup 1 --packed-steps 0"
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --brute-sample-mib 0 $extra 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); p=r.get('packed_text_shadow') or {}
        print('$w', 'value', round(r['value'],1), 'ms', round(r['ms_per_step'],3), 'kernel', round(r['roofline']['kernel_ms'],3), 'frac', round(r['roofline']['frac'],4), 'hits', r['hits'], 'verify', round(r['verify_ms_per_step'],4), 'packed', round(p.get('Gbases_per_s',0),1), p.get('hits_equal_to_unpacked'))
"
done
timeout -k 10 400 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline 2>gpurun_out/c5.err | tee gpurun_out/bench_c5.json | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); j=r['journaled_sequence_tree']
        print('c5 value', round(r['value'],1), 'ms', round(r['ms_per_step'],3), 'kernel', round(r['roofline']['kernel_ms'],3), 'frac', round(r['roofline']['frac'],4), 'hits', r['hits'], 'found', r['needles_found_on_their_haplotype'], 'sharing', round(j['sharing'],2), 'index_ms', round(j['index_ms_rank0'],1), 'verify', round(r['verify_ms_per_step'],3), 'fan', round(r['fanout_ms_per_step'],3), 'cand', r['candidates'], 'bands', r['bands_verified'], r.get('brute_force_engine'))
" || tail -5 gpurun_out/c5.err
