mkdir -p gpurun_out/r3t
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "conv or train_step" > gpurun_out/r3t/tests.log 2>&1; tail -3 gpurun_out/r3t/tests.log
RCF_OVERLAP_WGRAD=0 PROF_ROWS=12 bash tools/prof_step.sh r3t_fp32 fp32 4 > gpurun_out/r3t/fp32.txt 2>&1; grep "sum of kernel\|weight_pairs\|ms/step over" gpurun_out/r3t/fp32.txt | cut -c1-160
grep -c "weight_pairs" gpurun_out/prof_r3t_fp32/p_kernel_stats.csv; grep "weight_pairs" gpurun_out/prof_r3t_fp32/p_kernel_stats.csv | cut -c1-120
