#!/bin/bash
for sd in 2 3 4 6 8; do for pw in 1 2; do
  echo -n "SHADE_DEFER=$sd PRIM_WEIGHT=$pw  "
  VK_SHADE_DEFER=$sd VK_PRIM_WEIGHT=$pw python tools/experiments/perf_quick.py --wl C3,C4 --reps 2 --no-check | tail -1
done; done
