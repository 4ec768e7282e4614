#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
echo "== pytest gpu ==" | tee $OUT/progress.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.log
SPM_HIP_FILTER_DYN=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter or golden or sharding" > $OUT/pytest_var.log 2>&1 || { tail -30 $OUT/pytest_var.log; exit 1; }
echo "dynamic variant: $(tail -1 $OUT/pytest_var.log)" | tee -a $OUT/progress.log
rm -f $OUT/sweep6.log
for rep in 1 2; do
for DYN in 0 1; do for U in 8 4; do for TH in 1024 768 512 256; do for SPW in 8 32; do
  echo -n "rep=$rep DYN=$DYN U=$U TH=$TH SPW=$SPW : " | tee -a $OUT/sweep6.log
  SPM_HIP_FILTER_DYN=$DYN SPM_HIP_FILTER_U=$U SPM_HIP_FILTER_THREADS=$TH SPM_HIP_FILTER_SPANS_PER_WAVE=$SPW timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --brute-sample-mib 0 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(round(r['value'],1), round(r['ms_per_step'],3), round(r['roofline']['kernel_ms'],3), round(r['roofline']['frac'],4), r['hits'])
" | tee -a $OUT/sweep6.log
done; done; done; done; done
echo "== done ==" | tee -a $OUT/progress.log
