"""Copy the summaries of a scripts/gpu_session.sh run from gpurun_out/ into profiles/<round>/ (tracked)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

def newest(pattern):
    """gpurun merges into gpurun_out/ without deleting earlier runs: take the most recent match."""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]


tag = sys.argv[1] if len(sys.argv) > 1 else "v5"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
src, dst = "gpurun_out", os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
f = newest(f"{src}/prof_c3/*/*_kernel_stats.csv")[0]
shutil.copy(f, f"{dst}/c3_16GiB_kernel_stats_{tag}.csv")
for name in ("bench_c3", "bench_c2", "bench_c4", "bench_c5", "bench_2rank_gloo", "bench_c5_2rank_gloo"):
    if os.path.exists(f"{src}/{name}.json"):
        shutil.copy(f"{src}/{name}.json", f"{dst}/{name}_{tag}.json")
for f5 in newest(f"{src}/prof_c5/*/*_kernel_stats.csv"):
    shutil.copy(f5, f"{dst}/c5_kernel_stats_{tag}.csv")
out = {}
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = newest(f"{src}/{d}/*/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "spm_hip" in r["Kernel_Name"]]
    with open(f"{dst}/c3_16GiB_pmc_{name}_{tag}.csv", "w") as g:
        w = csv.writer(g)
        w.writerow(["kernel", "counter", "value_KB", "duration_ns"])
        for r in rows:
            w.writerow([r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Counter_Value"],
                        int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
    vals = [float(r["Counter_Value"]) for r in rows if "seed_filter" in r["Kernel_Name"]]
    out[name] = sum(vals) / len(vals)
traffic = out["FETCH_SIZE"] * 1024 * 2 + out["WRITE_SIZE"] * 1024
json.dump({"c3": {"text_bytes_per_gpu": 17179869184, "kernel": "seed_filter_kernel",
                  "FETCH_SIZE_KB": out["FETCH_SIZE"], "WRITE_SIZE_KB": out["WRITE_SIZE"],
                  "hbm_bytes_per_launch": traffic,
                  "method": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes "
                            f"(profiles/{rnd}/c3_16GiB_pmc_*_{tag}.csv); FETCH_SIZE doubled per MI355X_MICROARCH.md "
                            "(gfx950 tallies the 128-B requests of a 16 B/lane stream at 64 B), WRITE_SIZE as read"}},
          open("profiles/pmc_traffic.json", "w"), indent=1)
print("traffic/algorithmic =", traffic / 17179869184)
with open(f"{dst}/c3_16GiB_pmc_SQ_{tag}.csv", "w") as g:
    w = csv.writer(g)
    w.writerow(["kernel", "counter", "value", "duration_ns"])
    for d in ("pmc_sq", "pmc_sq2"):
        f = newest(f"{src}/{d}/*/*_counter_collection.csv")[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "seed_filter" in r["Kernel_Name"]:
                w.writerow([r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Counter_Value"],
                            int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(k, round(sum(v) / len(v)))
# brute engine: VALU wave-instructions per wave-step (= per lane-step) of the kernel that ran, 0.25 GiB slice
brute = {}
NB = 1 << 28
for d, wl, n_pat, key in (("pmc_brute_c3", "c3", 1024, "c3"), ("pmc_brute_c3_full", "c3", 1024, "c3_full_width"),
                          ("pmc_brute_c2", "c2", 1024, "c2")):
    fs = newest(f"{src}/{d}/*/*_counter_collection.csv")
    if not fs:
        continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "brute_kernel" in r["Kernel_Name"] or "cutoff_kernel" in r["Kernel_Name"]]
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "SQ_INSTS_VALU" not in agg:
        continue
    valu = sum(agg["SQ_INSTS_VALU"]) / len(agg["SQ_INSTS_VALU"])
    salu = sum(agg["SQ_INSTS_SALU"]) / len(agg["SQ_INSTS_SALU"])
    wave_steps = (n_pat / 64) * NB
    brute[key] = {"kernel": rows[0]["Kernel_Name"].split("(")[0].split("<")[0].replace("void spm_hip::", ""),
                  "needles": n_pat, "sample_bytes": NB, "SQ_INSTS_VALU": valu, "SQ_INSTS_SALU": salu,
                  "valu_per_lane_step": valu / wave_steps, "salu_per_wave_step": salu / wave_steps,
                  "method": "rocprofv3 --pmc SQ_INSTS_VALU over bench.py --engine brute --text-gib 0.25; wave "
                            "instructions / ((needles / 64) x text symbols); includes the tile warm-up columns"}
    with open(f"{dst}/brute_{key}_pmc_{tag}.csv", "w") as g:
        w = csv.writer(g)
        w.writerow(["kernel", "counter", "value", "duration_ns"])
        for r in rows:
            w.writerow([r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Counter_Value"],
                        int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
if brute:
    json.dump(brute, open("profiles/brute_valu.json", "w"), indent=1)
    print({k: round(v["valu_per_lane_step"], 2) for k, v in brute.items()})
if os.path.exists(f"{src}/valu_probe.jsonl"):
    shutil.copy(f"{src}/valu_probe.jsonl", f"{dst}/valu_probe_{tag}.jsonl")
print(open(f"{dst}/c3_16GiB_kernel_stats_{tag}.csv").read()[:600])
