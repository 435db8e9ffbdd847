mkdir -p gpurun_out/r3e
python tools/power_probe.py > gpurun_out/r3e/power.txt 2>&1; cat gpurun_out/r3e/power.txt
