#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
R=$GRAFT_REPO_ROOT
echo "== pytest gpu ==" | tee $OUT/progress.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.log
echo "== bench default (C3, 16 GiB) ==" | tee -a $OUT/progress.log
timeout -k 10 600 python bench.py 2>&1 | tee $OUT/bench_c3.json || exit 1
echo "== bench c2 ==" | tee -a $OUT/progress.log
timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline 2>&1 | tee $OUT/bench_c2.json || exit 1
echo "== sweep threads/span ==" | tee -a $OUT/progress.log
for span in 0 16 64 256; do
  echo "span=$span" | tee -a $OUT/sweep.log
  SPM_HIP_FILTER_SPAN=$span timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['verify_ms_per_step'], r['candidates'])
" | tee -a $OUT/sweep.log
done
echo "== rocprof kernel trace ==" | tee -a $OUT/progress.log
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c3 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --brute-sample-mib 64 > $R/$OUT/prof_c3.log 2>&1
echo "== rocprof pmc FETCH_SIZE ==" | tee -a $R/$OUT/progress.log
cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/pmc_fetch.log 2>&1
echo "== rocprof pmc WRITE_SIZE ==" | tee -a $R/$OUT/progress.log
cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/pmc_write.log 2>&1
cd $R
find $OUT -name "*.csv" | head -20
echo "== done ==" | tee -a $OUT/progress.log
