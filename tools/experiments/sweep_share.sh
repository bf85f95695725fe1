for e in "X=1" "VK_CHUNK_CAP=16" "VK_CHUNK_CAP=8" "VK_TILE_ORDER=0" "VK_NO_DUAL_LAUNCH=1"; do echo "== $e"; env $e timeout -k 10 200 python tools/experiments/partition_probe.py C2 8 | tail -1; done
