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
for cfg in "8 1 1" "8 0 0" "4 1 0"; do set -- $cfg
  SPM_HIP_FILTER_U=$1 SPM_HIP_FILTER_NT=$2 SPM_HIP_FILTER_HASH=$3 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter or golden or sharding" > $OUT/pytest_var.log 2>&1 || { tail -30 $OUT/pytest_var.log; exit 1; }
  echo "variant $cfg: $(tail -1 $OUT/pytest_var.log)" | tee -a $OUT/progress.log
done
echo "== sweep U/NT/HASH ==" | tee -a $OUT/progress.log
rm -f $OUT/sweep3.log
for U in 4 8; do for NT in 0 1; do for HV in 0 1; do
  echo -n "U=$U NT=$NT HASH=$HV : " | tee -a $OUT/sweep3.log
  SPM_HIP_FILTER_U=$U SPM_HIP_FILTER_NT=$NT SPM_HIP_FILTER_HASH=$HV timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --brute-sample-mib 0 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(round(r['value'],1), round(r['ms_per_step'],3), round(r['roofline']['kernel_ms'],3), round(r['roofline']['frac'],4), r['candidates'])
" | tee -a $OUT/sweep3.log
done; done; done
echo "== PMC diag (default variant) ==" | tee -a $OUT/progress.log
cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/pmc_sq.log 2>&1
cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/pmc_sq2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/pmc_sq2.log 2>&1
cd $R
echo "== done ==" | tee -a $OUT/progress.log
