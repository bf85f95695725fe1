#!/bin/bash
# kernel experiments: bench the alternate builds in vecchio_amd/lib/exp/*.so (see ffi.device_lib_path)
# usage (GPU box): bash tools/experiments/exp_bench.sh "C2 C4" [steps]
mkdir -p gpurun_out
WLS=${1:-C2}
STEPS=${2:-2}
for lib in default vecchio_amd/lib/exp/*.so; do
  for w in $WLS; do
    if [ "$lib" == "default" ]; then unset VK_DEVICE_LIB; else export VK_DEVICE_LIB=$PWD/$lib; fi
    timeout -k 10 400 python bench.py --steps $STEPS --warmup 1 --no-cpu --no-traffic --workload $w 2>>gpurun_out/exp.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['config']['workload'][:3], d['value'], d['ms_per_step'])" || exit 1
  done
done
