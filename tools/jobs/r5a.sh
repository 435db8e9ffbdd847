mkdir -p gpurun_out/r5a
timeout 300 python tools/cu_mask_probe.py > gpurun_out/r5a/cu_mask_probe.txt 2>&1; cat gpurun_out/r5a/cu_mask_probe.txt | cut -c1-330
