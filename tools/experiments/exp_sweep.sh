#!/bin/bash
# parameter sweeps of the diagnostic switches (GPU box): bash tools/experiments/exp_sweep.sh "<bench args>" "VAR=a,b,c" ...
ARGS=$1; shift
mkdir -p gpurun_out
run() { env "$@" timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-also --no-verify --no-traffic $ARGS 2>>gpurun_out/exp.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for spec in "$@"; do
  var=${spec%%=*}; vals=${spec#*=}
  for v in ${vals//,/ }; do run $var=$v || exit 1; done
done
