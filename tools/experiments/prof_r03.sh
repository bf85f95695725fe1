#!/bin/bash
# rocprofv3 runs for the round-3 profiles (run on the GPU box via gpurun from the repo root):
#   OUT=gpurun_out/prof_x WL="--workload C2" STEPS=2 WARMUP=1 bash tools/experiments/prof_r03.sh [pmc] [cache]
# Kernel trace + stats first; PMC counters in their own passes (never combined with other trace domains).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${OUT:-gpurun_out/prof_r03}
WL=${WL:-"--workload C2"}
STEPS=${STEPS:-2}
WARMUP=${WARMUP:-1}
TMO=${TMO:-300}
mkdir -p gpurun_out; rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps $STEPS --warmup $WARMUP --no-cpu --no-verify --no-traffic --no-also $WL"
echo "$ARGS" > $OUT/command.txt
timeout -k 10 $TMO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || exit 1
pass() { # name counters...
  local name=$1; shift
  timeout -k 10 $TMO rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $ARGS > $OUT/pmc_$name.log 2>&1
}
for a in "$@"; do
  if [ "$a" == "pmc" ]; then
    pass fetch FETCH_SIZE || exit 2
    pass write WRITE_SIZE || exit 3
    pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU || exit 4
    pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 5
  fi
  if [ "$a" == "cache" ]; then
    pass tcc TCC_HIT_sum TCC_MISS_sum || exit 6
    pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum || exit 7
    pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum || exit 8
    pass scratch SQ_INSTS_VMEM_WR SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS || true
  fi
done
grep "^{" $OUT/trace.log | tail -1 | cut -c1-200
