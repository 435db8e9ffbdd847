python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py tests/test_model_gpu.py tests/test_pipeline_gpu.py -q 2>&1 | tail -3
