mkdir -p gpurun_out/r3o
timeout 900 python -m pytest tests/test_bf16_gpu.py tests/test_kernels_gpu.py -x -q > gpurun_out/r3o/tests.log 2>&1; tail -3 gpurun_out/r3o/tests.log
timeout 900 python tools/ab_overlap.py 3 6 > gpurun_out/r3o/ab.txt 2>&1; tail -6 gpurun_out/r3o/ab.txt
python tools/layer_table.py bf16 > gpurun_out/r3o/layers_bf16.txt 2>&1; grep "====\|family totals" gpurun_out/r3o/layers_bf16.txt
