#!/bin/bash
# c3r at 1 % and 5 %: step time under a few scan-time knobs (resolve residency, band width, run length threshold)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python3 bench.py --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 --no-other-configs --workload c3r --steps 4 --warmup 3"
run() { # label, env...
  local label=$1; shift
  for f in 0.01 0.05; do
    env "$@" timeout -k 10 300 $B --repeat-frac $f --repeat-needle-every 16 2>/dev/null | tail -1 > gpurun_out/knob.json
    python3 -c "
import json; r=json.load(open('gpurun_out/knob.json'))
print('$label frac $f:', round(r['ms_per_step'],3), 'ms  verify', round(r['verify_ms_per_step'],3), 'bands', r['bands_verified'], 'hits', r['hits'], 'parity', r['parity_slice']['equal_to_brute_force_engine'])" | tee -a gpurun_out/c3r_knobs.log
  done
}
: > gpurun_out/c3r_knobs.log
run default A=1
run wgs6 SPM_HIP_RESOLVE_WGS_PER_CU=6
run wgs4 SPM_HIP_RESOLVE_WGS_PER_CU=4
run band64 SPM_HIP_FILTER_BAND=64
run band16 SPM_HIP_FILTER_BAND=16
run surv512 SPM_HIP_RESOLVE_SURV_PER_WG=512
run surv2048 SPM_HIP_RESOLVE_SURV_PER_WG=2048
run nopieces SPM_HIP_PIECES_CHECK=0
