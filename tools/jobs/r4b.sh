mkdir -p gpurun_out/r4b
python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py tests/test_abi_gpu.py -q -x 2>&1 | tail -3
python tools/ab_korder.py fp32 > gpurun_out/r4b/ab_korder_fp32.txt 2>&1; cat gpurun_out/r4b/ab_korder_fp32.txt | cut -c1-400
python tools/ab_step_knob.py korder 3 6 > gpurun_out/r4b/ab_step_korder.txt 2>&1; tail -4 gpurun_out/r4b/ab_step_korder.txt
for w in fwd dgrad; do bash tools/pmc_dispatch.sh fp32 $w > gpurun_out/r4b/pmcd_fp32_$w.txt 2>&1; cat gpurun_out/r4b/pmcd_fp32_$w.txt | cut -c1-260; done
