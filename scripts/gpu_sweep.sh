#!/bin/bash
# parity on the default + alternate filter variants, then a U/NT/HASH sweep of the C3 bench
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
echo "== pytest gpu ==" | tee $OUT/progress.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.log
for cfg in "8 1 1" "8 0 0" "4 1 0"; do set -- $cfg
  SPM_HIP_FILTER_U=$1 SPM_HIP_FILTER_NT=$2 SPM_HIP_FILTER_HASH=$3 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter or golden or sharding" > $OUT/pytest_var.log 2>&1 || { tail -30 $OUT/pytest_var.log; exit 1; }
  echo "variant $cfg: $(tail -1 $OUT/pytest_var.log)" | tee -a $OUT/progress.log
done
rm -f $OUT/sweep4.log
for U in 4 8; do for NT in 0 1; do for HV in 0 1; do for TH in 1024 512; do
  echo -n "U=$U NT=$NT HASH=$HV TH=$TH : " | tee -a $OUT/sweep4.log
  SPM_HIP_FILTER_U=$U SPM_HIP_FILTER_NT=$NT SPM_HIP_FILTER_HASH=$HV SPM_HIP_FILTER_THREADS=$TH timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --brute-sample-mib 0 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(round(r['value'],1), round(r['ms_per_step'],3), round(r['roofline']['kernel_ms'],3), round(r['roofline']['frac'],4), r['candidates'])
" | tee -a $OUT/sweep4.log
done; done; done; done
echo "== done ==" | tee -a $OUT/progress.log
