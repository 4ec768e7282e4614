"""Average the counters of rocprofv3 --pmc passes per kernel: one line per (kernel, counter).

    python scripts/summarise_pmc.py <pass dir> [<pass dir> ...]  >  profiles/rNN/<name>.txt
"""
import collections
import csv
import glob
import sys


def short(name):
    n = name.split("(")[0].replace("void spm_hip::", "").replace("spm_hip::", "")
    return n[:70]


for d in sys.argv[1:]:
    for f in sorted(glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(list)
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print(f"# {f}")
        for k in sorted(dur, key=lambda k: -sum(dur[k])):
            if sum(dur[k]) < 0.01 * sum(sum(v) for v in dur.values()):
                continue
            n = len(dur[k]) // max(1, len([c for (kk, c) in agg if kk == k]))
            print(f"{k}: launches {n}, avg_ns {sum(dur[k]) / len(dur[k]):.0f}")
            for (kk, c), v in sorted(agg.items()):
                if kk == k:
                    print(f"    {c:28s} {sum(v) / len(v):16.0f}")
