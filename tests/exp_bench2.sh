for lib in default vecchio_amd/lib/exp/*.so; do
  for w in C4 C2; do
    if [ "$lib" == "default" ]; then unset VK_DEVICE_LIB; else export VK_DEVICE_LIB=$PWD/$lib; fi
    for ord in 1 0; do
    VK_TILE_ORDER=$ord timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu --no-also --no-verify --workload $w 2>>gpurun_out/exp.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'order=$ord', d['config']['workload'][:3], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" || exit 1
    done
  done
done
