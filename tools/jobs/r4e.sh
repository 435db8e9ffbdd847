mkdir -p gpurun_out/r4e
python -m pytest tests/test_kernels_gpu.py -q -x -k "batchnorm" 2>&1 | tail -3
python tools/ab_step_knob.py bn_sweep 3 6 > gpurun_out/r4e/ab_step_bn_sweep.txt 2>&1; tail -6 gpurun_out/r4e/ab_step_bn_sweep.txt
