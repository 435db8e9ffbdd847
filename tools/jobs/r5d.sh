mkdir -p gpurun_out/r5d
timeout 900 python tools/ab_late_wgrad.py fp32 > gpurun_out/r5d/late_fp32.txt 2>&1; cat gpurun_out/r5d/late_fp32.txt | cut -c1-200
