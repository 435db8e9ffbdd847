python -m pytest tests/test_pipeline_gpu.py -q 2>&1 | tail -2
