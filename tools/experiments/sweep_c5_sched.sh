#!/bin/bash
# GPU box: scheduler thresholds on C5 (the near form).  usage: sweep_c5_sched.sh lib "defers" "weights"
for sd in $2; do for pw in $3; do
  echo -n "shade_defer $sd prim_weight $pw: "
  VK_SHADE_DEFER=$sd VK_PRIM_WEIGHT=$pw python tools/experiments/perf_quick.py --libs vecchio_amd/lib/exp/$1.so --wl C5 --reps 2 --no-check 2>&1 | grep -v "^+" | tail -1
done; done
