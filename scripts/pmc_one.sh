set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); OUT=$R/gpurun_out/pmc_c4_a; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --brute-sample-mib 0 --packed-steps 0 --no-other-configs"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAVES"
W=${W:-c4}
cd /tmp && timeout -k 10 400 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/pmc1_$W -- $B --workload $W --steps 1 --warmup 1 > $OUT/pmc1_$W.log 2>&1 || { tail -5 $OUT/pmc1_$W.log; exit 1; }
cd /tmp && timeout -k 10 400 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/pmc2_$W -- $B --workload $W --steps 1 --warmup 1 > $OUT/pmc2_$W.log 2>&1 || { tail -5 $OUT/pmc2_$W.log; exit 1; }
cd $R && python3 scripts/summarise_pmc.py $OUT/pmc1_$W $OUT/pmc2_$W > $OUT/${W}_pmc_SQ.txt; head -60 $OUT/${W}_pmc_SQ.txt
