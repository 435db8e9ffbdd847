mkdir -p gpurun_out/r4c
python tools/mall_probe.py > gpurun_out/r4c/mall_probe.txt 2>&1; cat gpurun_out/r4c/mall_probe.txt | cut -c1-250
python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -2
python tools/ab_korder.py fp32 > gpurun_out/r4c/ab_korder_fp32.txt 2>&1; cat gpurun_out/r4c/ab_korder_fp32.txt | cut -c1-300
bash tools/pmc_dispatch.sh fp32 fwd > gpurun_out/r4c/pmcd_fp32_fwd.txt 2>&1; head -6 gpurun_out/r4c/pmcd_fp32_fwd.txt | cut -c1-260
