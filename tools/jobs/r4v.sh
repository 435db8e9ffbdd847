python -m pytest tests/test_kernels_gpu.py -q -k "256x256 or permutation or exact_2x or cache_aware" 2>&1 | tail -3
grep "256x256 tile" gpurun_out/parity_report.txt | cut -c1-220
