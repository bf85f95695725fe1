export VK_DEVICE_LIB=$PWD/vecchio_amd/lib/exp/pool.so
echo "== parity with VK_POOL=6:112:1"
VK_POOL=6:112:1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 || exit 1
tools/experiments/pool_sweep2.sh "6:112:1 6:120:1 6:96:1 5:128:1 5:144:1 6:112:0" "48 64" "16 32"
