export VK_DEVICE_LIB=$PWD/vecchio_amd/lib/exp/pool3.so
echo "== parity with VK_POOL=6:80:1"
VK_POOL=6:80:1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 || exit 1
sed -i 's|exp/pool.so|exp/pool3.so|' tools/experiments/pool_sweep2.sh
tools/experiments/pool_sweep2.sh "6:80:1 6:72:1 5:96:1 5:88:1 4:120:1 6:80:0" "48 64" "16 32" C3,C4
