python tools/experiments/perf_quick.py --libs default --wl C2 --reps 3 2>&1 | grep -v "^+" | tail -1
VK_UNIT_FORM=0 python tools/experiments/perf_quick.py --libs vecchio_amd/lib/libvecchio_amd_debug.so --wl C2 --reps 3 2>&1 | grep -v "^+" | tail -1
VK_NO_LDS_SCENE=1 python tools/experiments/perf_quick.py --libs default --wl C2 --reps 3 2>&1 | grep -v "^+" | tail -1
python tools/experiments/perf_quick.py --libs vecchio_amd/lib/libvecchio_amd_debug.so --wl C2 --reps 3 2>&1 | grep -v "^+" | tail -1
