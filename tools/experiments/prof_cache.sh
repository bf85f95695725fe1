#!/bin/bash
# cache / TLB counters of the render kernel for a large scene (run on the GPU box from the repo root):
#   OUT=gpurun_out/prof_c5 ARGS="--workload C5 --spp 16" bash tools/experiments/prof_cache.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${OUT:-gpurun_out/prof_cache}
ARGS=${ARGS:-"--workload C5 --spp 8"}
rm -rf $OUT; mkdir -p $OUT
B="bench.py --steps 1 --warmup 0 --no-cpu $ARGS"
i=0
for ctrs in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 $B > $OUT/pmc_$i.log 2>&1 || exit $i
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    print({k: f"{v:.4g}" for k, v in agg.items()})
PY
grep "^{" $OUT/pmc_1.log | tail -1 | cut -c1-140
