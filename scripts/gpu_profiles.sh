#!/bin/bash
# Round profiles: kernel traces and PMC passes for every bench workload -> gpurun_out/prof_<tag>/ ; summaries are copied
# into profiles/<round>/ by scripts/collect_profiles.py.   gpurun --timeout 1200 -- 'TAG=r02 bash scripts/gpu_profiles.sh'
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); TAG=${TAG:-r02}; OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
say() { echo "== $* ==" | tee -a $OUT/progress.log; }
B="python3 $R/bench.py --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 --no-other-configs"
say "default bench line (all configs)"
timeout -k 10 600 python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
for W in c3 c2 c4 c5 c3r reads100; do
  extra="--steps 10 --warmup 3"; [ $W = c4 ] && extra="--steps 3 --warmup 1"; [ $W = reads100 ] && extra="--steps 3 --warmup 1"; [ $W = c3r ] && extra="--steps 5 --warmup 2"
  say "kernel trace $W"
  cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- $B --workload $W $extra > $OUT/trace_$W.log 2>&1 || { tail -5 $OUT/trace_$W.log; exit 1; }
done
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAVES"
for W in c3 c4 c5; do
  extra="--steps 3 --warmup 1"; [ $W = c4 ] && extra="--steps 1 --warmup 1"
  say "pmc SQ $W"
  cd /tmp && timeout -k 10 500 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/pmc1_$W -- $B --workload $W $extra > $OUT/pmc1_$W.log 2>&1 || { tail -5 $OUT/pmc1_$W.log; exit 1; }
  cd /tmp && timeout -k 10 500 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/pmc2_$W -- $B --workload $W $extra > $OUT/pmc2_$W.log 2>&1 || { tail -5 $OUT/pmc2_$W.log; exit 1; }
done
say "pmc HBM traffic c3"
cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_c3 -- $B --workload c3 --steps 3 --warmup 1 > $OUT/pmc_fetch_c3.log 2>&1 || { tail -5 $OUT/pmc_fetch_c3.log; exit 1; }
cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_c3 -- $B --workload c3 --steps 3 --warmup 1 > $OUT/pmc_write_c3.log 2>&1 || { tail -5 $OUT/pmc_write_c3.log; exit 1; }
say "repeat sweep (c3r at 0 / 0.1 / 1 / 5 %, needles across a stretch every 64th and every 8th)"
cd $R
for f in 0 0.001 0.01 0.05; do for e in 64 8; do
  timeout -k 10 400 python3 bench.py --workload c3r --repeat-frac $f --repeat-needle-every $e --steps 5 --warmup 2 --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 2>/dev/null | tail -1 > $OUT/c3r_f${f}_e${e}.json
  python3 -c "
import json; r=json.load(open('$OUT/c3r_f${f}_e${e}.json'))
print('c3r frac $f every $e:', round(r['value'],1), 'Gbases/s', round(r['ms_per_step'],3), 'ms  kernel', round(r['roofline']['kernel_ms'],3), 'cand', r['candidates'], 'bands', r['bands_verified'], 'hits', r['hits'], 'fallback_spans', r['fallback_spans'], 'parity', r['parity_slice']['equal_to_brute_force_engine'])" | tee -a $OUT/progress.log
done; done
say "hbm read probe + valu probe"
[ -x tools/hbm_read_probe ] && timeout -k 10 120 ./tools/hbm_read_probe > $OUT/hbm_read_probe.log 2>&1
say "done"
