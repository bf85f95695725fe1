#!/bin/bash
# rocprofv3 runs for the round-1 profiles (run on the GPU box via gpurun from the repo root):
#   OUT=gpurun_out/prof_x WL="--workload C2" bash tools/experiments/prof_r01.sh [pmc]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${OUT:-gpurun_out/prof_r01}
WL=${WL:-"--workload C2"}
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu $WL"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || exit 1
if [ "$1" == "pmc" ]; then
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_FLAT --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1 || exit 5
fi
grep "^{" $OUT/trace.log | tail -1 | cut -c1-200
