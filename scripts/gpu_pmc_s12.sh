#!/bin/bash
# PMC passes over the stride-1 (C4) and stride-2 (C5) seed-filter kernels: which unit binds them.
#   gpurun --timeout 1200 -- 'bash scripts/gpu_pmc_s12.sh'
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
say() { echo "== $* ==" | tee -a $OUT/progress.log; }
: > $OUT/progress.log
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  say "pytest -m gpu"
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
  tail -2 $OUT/pytest_gpu.log | tee -a $OUT/progress.log
fi
C4="python3 $R/bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 ${C4_EXTRA:-}"
C5="python3 $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --brute-sample-mib 0"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAVES"
for W in c4 c5; do
  [ $W = c4 ] && CMD="$C4" || CMD="$C5"
  say "kernel trace $W"
  rm -rf $OUT/s12_${W}_trace $OUT/s12_${W}_p1 $OUT/s12_${W}_p2
  cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s12_${W}_trace -- $CMD > $OUT/s12_${W}_trace.log 2>&1 || { tail -5 $OUT/s12_${W}_trace.log; exit 1; }
  say "pmc pass 1 $W"
  cd /tmp && timeout -k 10 500 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/s12_${W}_p1 -- $CMD > $OUT/s12_${W}_p1.log 2>&1 || { tail -5 $OUT/s12_${W}_p1.log; exit 1; }
  say "pmc pass 2 $W"
  cd /tmp && timeout -k 10 500 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/s12_${W}_p2 -- $CMD > $OUT/s12_${W}_p2.log 2>&1 || { tail -5 $OUT/s12_${W}_p2.log; exit 1; }
done
cd $R
python3 scripts/summarise_pmc.py $OUT/s12_c4_p1 $OUT/s12_c4_p2 > $OUT/s12_c4_summary.txt 2>&1
python3 scripts/summarise_pmc.py $OUT/s12_c5_p1 $OUT/s12_c5_p2 > $OUT/s12_c5_summary.txt 2>&1
cat $OUT/s12_c4_summary.txt $OUT/s12_c5_summary.txt
say "done"
