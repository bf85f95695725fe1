for cfg in "4 1" "5 1" "5 2" "6 2" "5 3" "8 2"; do set -- $cfg; echo -n "DEFER=$1 PW=$2 "; VK_SHADE_DEFER=$1 VK_PRIM_WEIGHT=$2 python tests/perf_quick.py --wl C2 --reps 4 --no-check | tail -1; done
for cfg in "5 1" "5 3"; do set -- $cfg; echo -n "DEFER=$1 PW=$2 "; VK_SHADE_DEFER=$1 VK_PRIM_WEIGHT=$2 python tests/perf_quick.py --wl C3,C4,C5 --reps 2 --no-check | tail -1; done
