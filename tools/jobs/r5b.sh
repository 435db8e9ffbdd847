mkdir -p gpurun_out/r5b
timeout 600 python tools/ab_late_wgrad.py fp32 > gpurun_out/r5b/late_fp32.txt 2>&1; cat gpurun_out/r5b/late_fp32.txt | cut -c1-200
timeout 600 python tools/ab_late_wgrad.py bf16 > gpurun_out/r5b/late_bf16.txt 2>&1; cat gpurun_out/r5b/late_bf16.txt | cut -c1-200
