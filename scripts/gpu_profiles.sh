#!/bin/bash
# Round profiles: kernel traces and PMC passes for every bench workload -> gpurun_out/prof_<tag>/ ; summaries are copied
# into profiles/<round>/ by scripts/collect_profiles.py.  Two calls (a gpurun call is limited to 20 minutes):
#   gpurun --timeout 1200 -- 'TAG=r03 PART=1 bash scripts/gpu_profiles.sh'     bench line, kernel traces, SQ counters
#   gpurun --timeout 1200 -- 'TAG=r03 PART=2 bash scripts/gpu_profiles.sh'     HBM traffic (FETCH_SIZE / WRITE_SIZE) of every config, repeat sweep, probes
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); TAG=${TAG:-r03}; PART=${PART:-1}; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
say() { echo "== $* ==" | tee -a $OUT/progress.log; }
B="python3 $R/bench.py --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 --no-other-configs"
steps_of() { case $1 in c4|reads100) echo "--steps 5 --warmup 3";; c3r) echo "--steps 5 --warmup 3";; *) echo "--steps 10 --warmup 3";; esac; }
if [ $PART = 1 ]; then
  say "default bench line (all configs)"
  timeout -k 10 600 python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
  for W in c3 c2 c4 c5 c3r reads100; do
    say "kernel trace $W"
    rm -rf $OUT/trace_$W
    cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- $B --workload $W $(steps_of $W) > $OUT/trace_$W.log 2>&1 || { tail -5 $OUT/trace_$W.log; exit 1; }
  done
  P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
  P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAVES"
  for W in c3 c4 c5 c3r; do
    extra="--steps 2 --warmup 2"
    [ $W = c3r ] && extra="--steps 2 --warmup 3 --repeat-frac 0.05"
    say "pmc SQ $W"
    rm -rf $OUT/pmc1_$W $OUT/pmc2_$W
    cd /tmp && timeout -k 10 500 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/pmc1_$W -- $B --workload $W $extra > $OUT/pmc1_$W.log 2>&1 || { tail -5 $OUT/pmc1_$W.log; exit 1; }
    cd /tmp && timeout -k 10 500 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/pmc2_$W -- $B --workload $W $extra > $OUT/pmc2_$W.log 2>&1 || { tail -5 $OUT/pmc2_$W.log; exit 1; }
  done
else
  for W in c3 c2 c4 c5 c3r reads100; do
    say "pmc HBM traffic $W"
    rm -rf $OUT/pmc_fetch_$W $OUT/pmc_write_$W
    cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$W -- $B --workload $W --steps 2 --warmup 2 > $OUT/pmc_fetch_$W.log 2>&1 || { tail -5 $OUT/pmc_fetch_$W.log; exit 1; }
    cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$W -- $B --workload $W --steps 2 --warmup 2 > $OUT/pmc_write_$W.log 2>&1 || { tail -5 $OUT/pmc_write_$W.log; exit 1; }
  done
  say "repeat sweep (c3r at 0 / 0.1 / 1 / 5 %, needles across a stretch every 128th / 64th / 16th / 8th)"
  cd $R
  for f in 0 0.001 0.01 0.05; do for e in 128 64 16 8; do
    timeout -k 10 400 python3 bench.py --workload c3r --repeat-frac $f --repeat-needle-every $e --steps 5 --warmup 3 --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 2>/dev/null | tail -1 > $OUT/c3r_f${f}_e${e}.json
    python3 -c "
import json; r=json.load(open('$OUT/c3r_f${f}_e${e}.json'))
print('c3r frac $f every $e:', round(r['value'],1), 'Gbases/s', round(r['ms_per_step'],3), 'ms  kernel', round(r['roofline']['kernel_ms'],3), 'cand', r['candidates'], 'bands', r['bands_verified'], 'hits', r['hits'], 'fallback_spans', r['fallback_spans'], 'parity', r['parity_slice']['equal_to_brute_force_engine'])" | tee -a $OUT/progress.log
  done; done
  say "rehearsal of the N > 1 path on this one GPU: 2 ranks, gloo (same rank-side code as RCCL: libspm_amd/dist.py)"
  for W in c3 c4 c5; do
    tg="--text-gib 2"; [ $W = c5 ] && tg=""
    SPM_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --workload $W $tg --steps 3 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 2>/dev/null | grep '^{' | tail -1 > $OUT/${W}_gpus2_gloo.json
    python3 -c "
import json; r=json.load(open('$OUT/${W}_gpus2_gloo.json')); print('$W x2 (gloo):', r['n_gpus'], 'ranks', round(r['ms_per_step'],3), 'ms/step', r['hits'], 'hits, all planted found', r['all_planted_found'])" | tee -a $OUT/progress.log
  done
  say "probes: HBM streaming read, L2 gathers"
  [ -x tools/hbm_read_probe ] && timeout -k 10 120 ./tools/hbm_read_probe > $OUT/hbm_read_probe.log 2>&1
  [ -x tools/l2_gather_probe ] && timeout -k 10 120 ./tools/l2_gather_probe 4096 > $OUT/l2_gather_probe.log 2>&1
fi
say "done part $PART"
