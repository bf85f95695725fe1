for cfg in "4 3" "3 3" "6 3" "4 2" "4 4" "6 4" "8 4"; do set -- $cfg; echo -n "C5 DEFER=$1 PW=$2 "; VK_SHADE_DEFER=$1 VK_PRIM_WEIGHT=$2 python tools/experiments/perf_quick.py --wl C5 --reps 2 --no-check | tail -1; done
for cc in 32 128; do echo -n "C2 CHUNK_CAP=$cc "; VK_CHUNK_CAP=$cc python tools/experiments/perf_quick.py --wl C2 --reps 3 --no-check | tail -1; done
