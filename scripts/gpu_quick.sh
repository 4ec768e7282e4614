#!/bin/bash
# quick GPU check: parity suite (or a subset: TESTS="tests/test_gpu_repeats.py") + selected bench lines (BENCH="c3 c3r")
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
if [ "${TESTS:-all}" != none ]; then
  T="${TESTS:-all}"; [ "$T" = all ] && T=tests
  timeout -k 10 900 python -m pytest $T -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1 || { tail -60 gpurun_out/pytest_gpu.log; exit 1; }
  tail -3 gpurun_out/pytest_gpu.log
fi
for w in ${BENCH:-}; do
  extra="--no-other-configs"
  [ $w = c4 ] && extra="$extra --steps 3 --warmup 1 --packed-steps 0"
  [ $w = reads100 ] && extra="$extra --steps 3 --warmup 1 --packed-steps 0"
  [ $w = c5 ] && extra="--steps 10 --warmup 2"
  [ $w = c3r ] && extra="$extra --steps 5 --warmup 2 --packed-steps 0"
  [ $w = c3r8 ] && { w=c3r; extra="$extra --steps 5 --warmup 2 --packed-steps 0 --repeat-needle-every 8"; }
  timeout -k 10 600 python bench.py --workload $w --no-cpu-baseline --brute-sample-mib 0 $extra ${BENCH_EXTRA:-} 2>gpurun_out/bench_$w.err | tee gpurun_out/bench_$w.json | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l)
        keep={k:r.get(k) for k in ('value','ms_per_step','hits','all_planted_found','verify_ms_per_step','candidates','bands_verified','fell_back','fallback_spans','parity_slice','needles_found_on_their_haplotype','fanout_ms_per_step')}
        keep['kernel_ms']=r['roofline']['kernel_ms']; keep['frac']=round(r['roofline']['frac'],4); keep['launches']=r['roofline'].get('launches_per_step')
        print('$w', json.dumps(keep))
" || { tail -5 gpurun_out/bench_$w.err; exit 1; }
done
