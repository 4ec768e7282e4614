"""VGPR / scratch use of the kernels in libspm_amd/csrc/spm_hip.gfx950.s (`make -C libspm_amd/csrc asm` first)."""
import re
import subprocess
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else "seed_filter|verify|cutoff|minim"
txt = open("libspm_amd/csrc/spm_hip.gfx950.s").read()
rows = []
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    rows.append((name, re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1),
                 re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (n, v, sc), d in zip(rows, names):
    d = d.replace("void spm_hip::", "").split("(")[0]
    if re.search(pat, d):
        print(f"{d[:90]:90s} vgpr {v:>4s} scratch {sc}")
