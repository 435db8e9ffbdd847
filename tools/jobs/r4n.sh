mkdir -p gpurun_out/r4n
RCF_WGRAD_BIG=3 python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py -q -k "wgrad or conv_fwd_dgrad or region or pairs or conv_bf16" 2>&1 | tail -3
python tools/ab_wgrad_big.py > gpurun_out/r4n/ab_wgrad_big.txt 2>&1; cat gpurun_out/r4n/ab_wgrad_big.txt | cut -c1-330
python tools/ab_step_knob.py wgrad_big 3 6 > gpurun_out/r4n/ab_step.txt 2>&1; grep "step" gpurun_out/r4n/ab_step.txt
