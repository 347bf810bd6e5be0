mkdir -p gpurun_out/gw
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/gw/tests.txt
rm -f gpurun_out/gw/res.txt
for v in 0 1; do
  echo "== RST_RESBLOCK_STREAM=$v" >> gpurun_out/gw/res.txt
  RST_RESBLOCK_STREAM=$v timeout 300 python tools/bench_kernels.py res64 res64pre res64post res128 conv:64x240000x64x128x8x4:elu conv:64x12000x256x128x3x1:elu 2>&1 | grep -v amdgpu.ids >> gpurun_out/gw/res.txt
done
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/gw/bench_r1.json 2> gpurun_out/gw/bench_r1.err
