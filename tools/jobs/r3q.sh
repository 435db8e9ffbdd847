mkdir -p gpurun_out/r3q
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py tests/test_model_gpu.py tests/test_vit_gpu.py -x -q > gpurun_out/r3q/tests.log 2>&1; tail -3 gpurun_out/r3q/tests.log
python tools/layer_table.py fp32 > gpurun_out/r3q/layers_fp32.txt 2>&1; grep "====\|family totals" gpurun_out/r3q/layers_fp32.txt
python tools/layer_table.py bf16 > gpurun_out/r3q/layers_bf16.txt 2>&1; grep "====\|family totals" gpurun_out/r3q/layers_bf16.txt
