#!/bin/bash
# Why is the streaming kernel 2.6 ms in some launches and 2.9 ms in others on the repeat text?  Counters per launch
# (the first launch of a process is fast, later ones slow): cycles vs duration says whether the clock moved.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); OUT=$R/gpurun_out/bimodal; mkdir -p $OUT; export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 --no-other-configs --workload c3r --steps 6 --warmup 3"
cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1
cd /tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum --kernel-trace --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    fs = sorted(glob.glob(f"gpurun_out/bimodal/{p}/**/*counter_collection.csv", recursive=True))
    if not fs:
        print(p, "no counters"); continue
    rows = [r for r in csv.DictReader(open(fs[-1])) if "seed_filter_kernel" in r["Kernel_Name"]]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(r["Dispatch_Id"], {"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})[r["Counter_Name"]] = float(r["Counter_Value"])
    for d, v in by.items():
        print(p, d, {k: (round(x, 3) if k == "ms" else f"{x:.4g}") for k, x in v.items()})
PY
