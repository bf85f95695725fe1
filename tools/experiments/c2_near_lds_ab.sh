# round 5: the near form of exact re-treeing on the InOneWeekend scene staged in LDS (the default since its reach spans the field)
# against the unit form (VK_NEAR_FIRST=0, debug library) and against itself from global memory (VK_NEAR_LDS=0)
D=vecchio_amd/lib/libvecchio_amd_debug.so
python tools/experiments/perf_quick.py --libs default --wl C2 --reps 3 2>&1 | grep -v "^+" | tail -1
VK_NEAR_FIRST=0 python tools/experiments/perf_quick.py --libs $D --wl C2 --reps 3 2>&1 | grep -v "^+" | tail -1
VK_NEAR_LDS=0 python tools/experiments/perf_quick.py --libs default --wl C2 --reps 3 2>&1 | grep -v "^+" | tail -1
python tools/experiments/perf_quick.py --libs default --wl C2 --reps 3 2>&1 | grep -v "^+" | tail -1
VK_NEAR_FIRST=0 python tools/experiments/perf_quick.py --libs $D --wl C2 --reps 3 2>&1 | grep -v "^+" | tail -1
python tools/experiments/perf_quick.py --libs default --wl C5 --reps 2 2>&1 | grep -v "^+" | tail -1
