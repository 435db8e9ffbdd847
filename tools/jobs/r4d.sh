mkdir -p gpurun_out/r4d
python -m pytest tests/test_kernels_gpu.py -q -x -k "batchnorm" 2>&1 | tail -3
python tools/ab_step_knob.py bn_sweep 3 6 > gpurun_out/r4d/ab_step_bn_sweep.txt 2>&1; tail -4 gpurun_out/r4d/ab_step_bn_sweep.txt
RCF_BN_SWEEP=0 PROF_ROWS=12 bash tools/prof_step.sh sw0 fp32 4 > gpurun_out/r4d/prof_sw0.txt 2>&1; cut -c1-150 gpurun_out/r4d/prof_sw0.txt | grep -i "bn_\|colreduce2\|sum of"
RCF_BN_SWEEP=1 PROF_ROWS=12 bash tools/prof_step.sh sw1 fp32 4 > gpurun_out/r4d/prof_sw1.txt 2>&1; cut -c1-150 gpurun_out/r4d/prof_sw1.txt | grep -i "bn_\|colreduce2\|sum of"
