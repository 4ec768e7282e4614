#!/bin/bash
# One MI355X session: parity tests, C++ API cases, smoke, bench (C3 default, C2, C4), rocprofv3 kernel trace and the
# PMC passes whose summaries go to profiles/.  Usage (from the repo root, through gpurun):
#   gpurun --timeout 1200 -- 'bash scripts/gpu_session.sh'
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
say() { echo "== $* ==" | tee -a $OUT/progress.log; }
: > $OUT/progress.log
say "pytest -m gpu"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log | tee -a $OUT/progress.log
say "C++ reference cases / jst cases"
timeout -k 10 300 ./tests/cpp/reference_cases 2>&1 | tail -2 | tee -a $OUT/progress.log
timeout -k 10 300 ./tests/cpp/jst_cases 2>&1 | tail -9 | tee -a $OUT/progress.log
say "smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/progress.log
say "bench C3 (default)"
timeout -k 10 600 python bench.py 2>/dev/null | tee $OUT/bench_c3.json | cut -c1-300
say "bench C2 / C4 / 2-rank rehearsal"
timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_c2.json | cut -c1-200
timeout -k 10 600 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 2>/dev/null | tee $OUT/bench_c4.json | cut -c1-200
timeout -k 10 400 python bench.py --workload c5 --steps 10 --warmup 2 2>/dev/null | tee $OUT/bench_c5.json | cut -c1-200
SPM_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29615 bench.py --gpus 2 --workload c5 --steps 3 --warmup 1 --text-gib 0.03125 --no-cpu-baseline 2>/dev/null | tail -1 | tee $OUT/bench_c5_2rank_gloo.json | cut -c1-200
SPM_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 3 --warmup 1 --text-gib 2 2>/dev/null | tail -1 | tee $OUT/bench_2rank_gloo.json | cut -c1-200
say "rocprofv3 kernel trace + PMC passes (C3)"
rm -rf $OUT/prof_c3 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_sq2
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0"
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c3 -- $B > $R/$OUT/prof_c3.log 2>&1
B3="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0"
cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_fetch -- $B3 > $R/$OUT/pmc_fetch.log 2>&1
cd /tmp && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_write -- $B3 > $R/$OUT/pmc_write.log 2>&1
cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$OUT/pmc_sq -- $B3 > $R/$OUT/pmc_sq.log 2>&1
cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $R/$OUT/pmc_sq2 -- $B3 > $R/$OUT/pmc_sq2.log 2>&1
rm -rf $OUT/prof_c5
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c5 -- python3 $R/bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/prof_c5.log 2>&1
cd $R
say "brute engine: VALU instructions per lane-step (PMC) + VALU issue peak"
rm -rf $OUT/pmc_brute_c3 $OUT/pmc_brute_c3_full $OUT/pmc_brute_c2
BB="--engine brute --text-gib 0.25 --steps 2 --warmup 1 --no-cpu-baseline --packed-steps 0"
cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/pmc_brute_c3 -- python3 $R/bench.py $BB > $R/$OUT/pmc_brute_c3.log 2>&1
export SPM_HIP_BRUTE_CUTOFF=0
cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/pmc_brute_c3_full -- python3 $R/bench.py $BB > $R/$OUT/pmc_brute_c3_full.log 2>&1
unset SPM_HIP_BRUTE_CUTOFF
cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/pmc_brute_c2 -- python3 $R/bench.py --workload c2 $BB > $R/$OUT/pmc_brute_c2.log 2>&1
cd $R
[ -x tools/valu_probe ] || (cd tools && hipcc -O3 --offload-arch=gfx950 -o valu_probe valu_probe.hip)
timeout -k 10 120 ./tools/valu_probe > $OUT/valu_probe.jsonl
tail -3 $OUT/valu_probe.jsonl
say "done"
