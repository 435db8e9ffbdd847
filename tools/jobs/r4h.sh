mkdir -p gpurun_out/r4h
python -m pytest tests/test_bf16_gpu.py tests/test_flowhead_gpu.py -q 2>&1 | tail -4
grep -n "bf16 step vs reference \[small\]" gpurun_out/parity_report.txt | cut -c1-600
