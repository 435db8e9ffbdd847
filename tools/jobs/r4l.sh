mkdir -p gpurun_out/r4l
python -m pytest tests/test_kernels_gpu.py -q -k "resize or column_tile" 2>&1 | tail -3
python tools/bench_resize.py > gpurun_out/r4l/bench_resize.txt 2>&1; cat gpurun_out/r4l/bench_resize.txt | cut -c1-250
