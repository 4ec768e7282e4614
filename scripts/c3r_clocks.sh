#!/bin/bash
# Does the streaming kernel's slow mode coincide with a clock or power state?  rocm-smi sampled in the background while
# bench.py scans the repeat text; the per-scan kernel times come from SPM_HIP_TRACE.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); OUT=$R/gpurun_out/clocks; mkdir -p $OUT
B="python3 $R/bench.py --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 --no-other-configs --workload c3r --steps 40 --warmup 2 --repeat-frac 0.01 --repeat-needle-every 8"
for i in 1 2 3; do
  ( while true; do date +%s.%N; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (junction|memory)" ; sleep 0.1; done ) > $OUT/smi_$i.log 2>&1 &
  SMI=$!
  SPM_HIP_TRACE=1 $B 2>&1 >/dev/null | grep "scan \[0, 1717" | sed -e "s/.*(main \([0-9.]*\),.*/\1/" | tr "\n" " " > $OUT/kern_$i.txt
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  echo "process $i kernel ms: $(cat $OUT/kern_$i.txt)"
  echo "  distinct smi lines:"; grep -v "^[0-9]*\.[0-9]*$" $OUT/smi_$i.log | sed -e "s/^GPU\[[0-9]*\][ \t]*: //" | sort | uniq -c | sort -rn | head -14
done
