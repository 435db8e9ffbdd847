mkdir -p gpurun_out/r3r
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py tests/test_model_gpu.py tests/test_vit_gpu.py tests/test_stage2_gpu.py tests/test_pipeline_gpu.py -x -q > gpurun_out/r3r/tests.log 2>&1; tail -3 gpurun_out/r3r/tests.log
timeout 900 python tools/ab_overlap.py 2 6 > gpurun_out/r3r/ab.txt 2>&1; tail -6 gpurun_out/r3r/ab.txt
