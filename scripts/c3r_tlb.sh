#!/bin/bash
# The streaming kernel's slow mode on the repeat text is decided per process: translation (UTCL1) counters per launch over a
# few processes, to see whether the slow ones miss more in the TLB.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); OUT=$R/gpurun_out/tlb; mkdir -p $OUT; export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 --no-other-configs --workload c3r --steps 3 --warmup 2 --repeat-frac 0.01 --repeat-needle-every 8"
for i in 1 2 3 4 5; do
  rm -rf $OUT/p$i
  (cd /tmp && rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --kernel-trace --output-format csv -d $OUT/p$i -- $B > $OUT/p$i.log 2>&1)
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for i in range(1, 6):
    fs = sorted(glob.glob(f"gpurun_out/tlb/p{i}/**/*counter_collection.csv", recursive=True))
    if not fs:
        print("process", i, "no counters (", open(f"gpurun_out/tlb/p{i}.log").read()[-300:], ")"); continue
    rows = [r for r in csv.DictReader(open(fs[-1])) if "seed_filter_kernel" in r["Kernel_Name"]]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(r["Dispatch_Id"], {"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})[r["Counter_Name"]] = float(r["Counter_Value"])
    for d, v in by.items():
        if v["ms"] > 1:
            print("process", i, "dispatch", d, {k: (round(x, 3) if k == "ms" else f"{x:.4g}") for k, x in v.items()})
PY
