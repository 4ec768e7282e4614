#!/bin/bash
# kernel breakdown of the repeat-rich workload at a given repeat fraction: rocprofv3 --kernel-trace --stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); export TMPDIR=/tmp
F=${F:-0.05}; E=${E:-64}; OUT=$R/gpurun_out/c3r_f${F}_e${E}; rm -rf $OUT; mkdir -p $OUT
B="python3 $R/bench.py --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 --no-other-configs --workload c3r --repeat-frac $F --repeat-needle-every $E --steps 3 --warmup 2"
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
cd $R && python3 - <<PY
import csv, glob, collections
t=sorted(glob.glob("$OUT/trace/**/*_kernel_trace.csv", recursive=True))[-1]
calls=collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    calls[r["Kernel_Name"].split("(")[0].replace("void spm_hip::","")].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
with open("$OUT/kernel_calls.txt","w") as g:
    g.write("# c3r repeat fraction $F, every ${E}th needle across a stretch: duration (ms) of every call of each kernel (first calls include warm-up)\n")
    for k,v in sorted(calls.items(), key=lambda kv:-sum(kv[1])):
        if sum(v)>0.02: g.write(f"{k[:70]:70s} n={len(v):3d}  "+" ".join(f"{x:.3f}" for x in v[-8:])+"\n")
print(open("$OUT/kernel_calls.txt").read())
PY
tail -1 $OUT/trace.log | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('c3r', r['value'], 'Gbases/s', r['ms_per_step'], 'ms; kernel', r['roofline']['kernel_ms'], 'cand', r['candidates'], 'bands', r['bands_verified'], 'hits', r['hits'], 'fallback', r['fallback_spans'])"
