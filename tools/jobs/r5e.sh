mkdir -p gpurun_out/r5e
python -m pytest tests/test_pipeline_gpu.py -q -k second_stream 2>&1 | tail -2
timeout 600 python tools/ab_late_wgrad.py bf16 > gpurun_out/r5e/late_bf16.txt 2>&1; cat gpurun_out/r5e/late_bf16.txt | cut -c1-200
