#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
rm -f $OUT/sweep_packed.log
for TH in 512 768 1024; do for SPW in 8 32 128; do
  echo -n "TH=$TH SPW=$SPW : " | tee -a $OUT/sweep_packed.log
  SPM_HIP_FILTER_THREADS=$TH SPM_HIP_FILTER_SPANS_PER_WAVE=$SPW timeout -k 10 200 python bench.py --steps 3 --warmup 1 --packed-steps 20 --no-cpu-baseline --brute-sample-mib 0 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); p=r['packed_text_shadow']; print(round(p['Gbases_per_s'],1), round(p['ms_per_step'],3), round(p['kernel_ms'],3), p['hits_equal_to_unpacked'])
" | tee -a $OUT/sweep_packed.log
done; done
