#!/bin/bash
# rocprofv3 kernel trace of bench.py workloads: W="c3 c3r" [ARGS="..."] bash scripts/gpu_trace.sh -> gpurun_out/trace_<w>_stats.csv
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for w in ${W:-c3}; do
  rm -rf $OUT/trace_$w
  extra="--steps 5 --warmup 2"; [ $w = c4 ] && extra="--steps 2 --warmup 1"
  cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 --no-other-configs $extra ${ARGS:-} > $OUT/trace_$w.log 2>&1 || { tail -5 $OUT/trace_$w.log; exit 1; }
  f=$(ls $OUT/trace_$w/*/*_kernel_stats.csv | tail -1); cp $f $OUT/trace_${w}_stats.csv
  echo "== $w"; python3 $R/scripts/trace_tail.py $OUT/trace_$w 9
done
