#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== cpp reference cases ==" | tee $OUT/progress.log
timeout -k 10 120 ./tests/cpp/reference_cases 2>&1 | tee -a $OUT/progress.log
echo "== pytest gpu ==" | tee -a $OUT/progress.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.log
echo "== 2-rank rehearsal (gloo, both ranks on cuda:0) ==" | tee -a $OUT/progress.log
SPM_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --text-gib 2 2>&1 | tail -3 | tee $OUT/bench_2rank_gloo.json
echo "== bench default ==" | tee -a $OUT/progress.log
timeout -k 10 600 python bench.py 2>&1 | tee $OUT/bench_c3.json || exit 1
echo "== rocprof kernel trace (clean: no small-sample scans) ==" | tee -a $OUT/progress.log
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c3 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/prof_c3.log 2>&1
cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/pmc_fetch.log 2>&1
cd /tmp && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --brute-sample-mib 0 > $R/$OUT/pmc_write.log 2>&1
cd $R
echo "== done ==" | tee -a $OUT/progress.log
