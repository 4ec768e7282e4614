#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
echo "== pytest gpu (default = CHD fingerprints) ==" | tee $OUT/progress.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.log
SPM_HIP_FILTER_HASH=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter or golden or sharding" > $OUT/pytest_var.log 2>&1 || { tail -30 $OUT/pytest_var.log; exit 1; }
echo "bloom variant: $(tail -1 $OUT/pytest_var.log)" | tee -a $OUT/progress.log
rm -f $OUT/sweep5.log
for HV in 2 1; do for U in 8 4; do for TH in 1024 512; do
  echo -n "HASH=$HV U=$U TH=$TH : " | tee -a $OUT/sweep5.log
  SPM_HIP_FILTER_U=$U SPM_HIP_FILTER_HASH=$HV SPM_HIP_FILTER_THREADS=$TH timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --brute-sample-mib 0 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(round(r['value'],1), round(r['ms_per_step'],3), round(r['roofline']['kernel_ms'],3), round(r['roofline']['frac'],4), r['candidates'], r['hits'])
" | tee -a $OUT/sweep5.log
done; done; done
echo "== done ==" | tee -a $OUT/progress.log
