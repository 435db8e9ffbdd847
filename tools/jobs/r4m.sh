mkdir -p gpurun_out/r4m
RCF_WGRAD_BIG=1 python -m pytest tests/test_kernels_gpu.py -q -k "wgrad or conv_fwd_dgrad or region or pairs" 2>&1 | tail -3
python tools/ab_wgrad_big.py > gpurun_out/r4m/ab_wgrad_big.txt 2>&1; cat gpurun_out/r4m/ab_wgrad_big.txt | cut -c1-250
python tools/ab_step_knob.py wgrad_big 3 6 > gpurun_out/r4m/ab_step.txt 2>&1; grep "fp32 step" gpurun_out/r4m/ab_step.txt
