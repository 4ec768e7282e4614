#!/bin/bash
# rocprofv3 --kernel-trace of one bench workload: per-kernel durations of the last calls -> gpurun_out/trace_<tag>/kernel_calls.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); export TMPDIR=/tmp
W=${W:-c5}; TAG=${TAG:-$W}; OUT=$R/gpurun_out/trace_$TAG; rm -rf $OUT; mkdir -p $OUT
B="python3 $R/bench.py --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 --no-other-configs --workload $W --steps ${STEPS:-5} --warmup 2 $EXTRA"
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
cd $R && python3 - <<PY
import csv, glob, collections
t=sorted(glob.glob("$OUT/trace/**/*_kernel_trace.csv", recursive=True))[-1]
calls=collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    calls[r["Kernel_Name"].split("(")[0].replace("void spm_hip::","")].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
with open("$OUT/kernel_calls.txt","w") as g:
    g.write("# $W $EXTRA: duration (ms) of every call of each kernel, in launch order (the first calls include warm-up)\n")
    for k,v in sorted(calls.items(), key=lambda kv:-sum(kv[1])):
        if sum(v)>0.01: g.write(f"{k[:70]:70s} n={len(v):3d}  "+" ".join(f"{x:.3f}" for x in v[-8:])+"\n")
print(open("$OUT/kernel_calls.txt").read())
PY
tail -1 $OUT/trace.log | cut -c1-400
