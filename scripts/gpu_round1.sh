#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (default = C3 at 16 GiB), brute reference, rocprof trace, sweep.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
echo "== pytest gpu ==" | tee $OUT/progress.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.log
echo "== smoke ==" | tee -a $OUT/progress.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee -a $OUT/progress.log || exit 1
echo "== bench 1 GiB quick ==" | tee -a $OUT/progress.log
timeout -k 10 300 python bench.py --text-gib 1 --steps 5 --warmup 2 --no-cpu-baseline --brute-sample-mib 64 2>&1 | tee $OUT/bench_1g.json || exit 1
echo "== bench default (C3, 16 GiB) ==" | tee -a $OUT/progress.log
timeout -k 10 600 python bench.py 2>&1 | tee $OUT/bench_c3.json || exit 1
echo "== sweep ==" | tee -a $OUT/progress.log
for probes in 2 3 4; do for thr in 256 512 1024; do
  echo "probes=$probes threads=$thr" | tee -a $OUT/sweep.log
  SPM_HIP_FILTER_PROBES=$probes SPM_HIP_FILTER_THREADS=$thr timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['verify_ms_per_step'], r['candidates'])
" | tee -a $OUT/sweep.log
done; done
echo "== rocprof kernel trace ==" | tee -a $OUT/progress.log
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_c3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --brute-sample-mib 64 > $GRAFT_REPO_ROOT/$OUT/prof_c3.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof_c3 -name "*stats*" | head; 
echo "== done ==" | tee -a $OUT/progress.log
