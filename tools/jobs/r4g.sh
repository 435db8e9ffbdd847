mkdir -p gpurun_out/r4g
python -m pytest tests/test_flowhead_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_bf16_gpu.py -q 2>&1 | tail -4
