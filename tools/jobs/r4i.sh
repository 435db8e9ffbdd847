mkdir -p gpurun_out/r4i
python tools/power_probe.py > gpurun_out/r4i/power_probe.txt 2>&1; cat gpurun_out/r4i/power_probe.txt | cut -c1-330
