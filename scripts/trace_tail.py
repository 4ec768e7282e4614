"""Durations of the last calls of each kernel in a rocprofv3 kernel trace: python scripts/trace_tail.py <trace dir> [n]"""
import collections, csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = sorted(glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True))[-1]
calls = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void spm_hip::", "").replace("spm_hip::", "")[:60]
    calls[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(calls.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 0.05:
        print(f"{k:60s} calls {len(v):4d}  last: " + " ".join(f"{x:.3f}" for x in v[-n:]))
