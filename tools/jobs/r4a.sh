mkdir -p gpurun_out/r4a
python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py -q -x 2>&1 | tail -3
python tools/ab_korder.py bf16 > gpurun_out/r4a/ab_korder_bf16.txt 2>&1; cat gpurun_out/r4a/ab_korder_bf16.txt | cut -c1-400
python tools/ab_wgrad_xcd.py > gpurun_out/r4a/ab_wgrad_xcd.txt 2>&1; cat gpurun_out/r4a/ab_wgrad_xcd.txt | cut -c1-330
python tools/ab_step_knob.py korder 3 6 > gpurun_out/r4a/ab_step_korder.txt 2>&1; tail -4 gpurun_out/r4a/ab_step_korder.txt
for w in fwd dgrad wgrad; do bash tools/pmc_dispatch.sh bf16 $w > gpurun_out/r4a/pmcd_bf16_$w.txt 2>&1; cat gpurun_out/r4a/pmcd_bf16_$w.txt | cut -c1-260; done
bash tools/pmc_dispatch.sh fp32 wgrad > gpurun_out/r4a/pmcd_fp32_wgrad.txt 2>&1; cat gpurun_out/r4a/pmcd_fp32_wgrad.txt | cut -c1-260
