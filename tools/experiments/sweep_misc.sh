echo -n "C2 NO_LDS_SCENE "; VK_NO_LDS_SCENE=1 python tools/experiments/perf_quick.py --wl C2 --reps 3 --no-check | tail -1
for v in "" "VK_TILE_ORDER=1"; do echo -n "C2 full $v "; env $v python bench.py --no-cpu --no-also --no-traffic --no-verify --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
