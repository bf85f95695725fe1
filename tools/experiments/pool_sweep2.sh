#!/bin/bash
# GPU box: throughput of the path pool per shape x thresholds (no parity run).  usage: pool_sweep2.sh "shapes" "threshs" "xmins" [wl]
export VK_DEVICE_LIB=$PWD/vecchio_amd/lib/exp/pool.so
wl=${4:-C3,C4}
echo "== baseline (VK_POOL=0)"
VK_POOL=0 timeout -k 10 300 python tools/experiments/perf_quick.py --libs $VK_DEVICE_LIB --wl $wl --reps 3 --no-check 2>&1 | grep -v "^+" | tail -1
for sh in $1; do for th in $2; do for xm in $3; do
    echo -n "VK_POOL=$sh thresh $th xmin $xm: "
    VK_POOL=$sh VK_POOL_THRESH=$th VK_POOL_XMIN=$xm VK_SHADE_DEFER=${SD:-0} timeout -k 10 300 python tools/experiments/perf_quick.py --libs $VK_DEVICE_LIB --wl $wl --reps 3 --no-check 2>&1 | grep -v "^+" | tail -1 || exit 1
done; done; done
