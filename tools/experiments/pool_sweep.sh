#!/bin/bash
# GPU box: parity of the path pool (per-sample tests through the experimental library) and its throughput per shape.
# usage: tools/experiments/pool_sweep.sh "5:80:1 4:104:1 ..." [wl]
export VK_DEVICE_LIB=$PWD/vecchio_amd/lib/exp/pool.so
shapes=${1:-"5:80:1"}; wl=${2:-C3,C4}
first=$(echo $shapes | cut -d' ' -f1)
echo "== parity with VK_POOL=$first"
VK_POOL=$first timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 || exit 1
echo "== baseline (VK_POOL=0)"
VK_POOL=0 timeout -k 10 300 python tools/experiments/perf_quick.py --libs $VK_DEVICE_LIB --wl $wl --reps 2 2>&1 | grep -v "^+" | tail -3
for sh in $shapes; do
  for th in ${THRESH:-64}; do for xm in ${XMIN:-8}; do
    echo "== VK_POOL=$sh thresh $th xmin $xm"
    VK_POOL=$sh VK_POOL_THRESH=$th VK_POOL_XMIN=$xm timeout -k 10 300 python tools/experiments/perf_quick.py --libs $VK_DEVICE_LIB --wl $wl --reps 2 2>&1 | grep -v "^+" | tail -2 || exit 1
  done; done
done
