python tools/absmax_trace.py fp32 2>&1 | tail -9 | cut -c1-200
python -m pytest tests/test_model_gpu.py tests/test_pipeline_gpu.py -q 2>&1 | tail -2
