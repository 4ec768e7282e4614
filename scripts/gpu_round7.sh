#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f $OUT/sweep7.log
for TH in 384 512 640; do for SPW in 16 32 64 128; do
  echo -n "TH=$TH SPW=$SPW : " | tee -a $OUT/sweep7.log
  SPM_HIP_FILTER_THREADS=$TH SPM_HIP_FILTER_SPANS_PER_WAVE=$SPW timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --brute-sample-mib 0 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(round(r['value'],1), round(r['ms_per_step'],3), round(r['roofline']['kernel_ms'],3), round(r['roofline']['frac'],4), r['hits'])
" | tee -a $OUT/sweep7.log
done; done
echo "== bench default ==" | tee -a $OUT/progress.log
timeout -k 10 600 python bench.py 2>&1 | tee $OUT/bench_c3.json || exit 1
timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline 2>&1 | tee $OUT/bench_c2.json || exit 1
echo "== rocprof ==" | tee -a $OUT/progress.log
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c3 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/prof_c3.log 2>&1
cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/pmc_fetch.log 2>&1
cd /tmp && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/pmc_write.log 2>&1
cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/pmc_sq.log 2>&1
cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $R/$OUT/pmc_sq2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/pmc_sq2.log 2>&1
cd $R; echo "== done ==" | tee -a $OUT/progress.log
