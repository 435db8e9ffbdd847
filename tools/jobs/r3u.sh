mkdir -p gpurun_out/r3u
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_dist_gpu.py tests/test_stage2_gpu.py -x -q > gpurun_out/r3u/tests.log 2>&1; tail -12 gpurun_out/r3u/tests.log | cut -c1-300
RCF_OVERLAP_WGRAD=0 PROF_ROWS=10 bash tools/prof_step.sh r3u_fp32 fp32 4 > gpurun_out/r3u/fp32.txt 2>&1; grep "ms/step over\|sum of kernel\|wprep" gpurun_out/r3u/fp32.txt | cut -c1-160
grep "wprep\|weight_pairs\|absmax" gpurun_out/prof_r3u_fp32/p_kernel_stats.csv | cut -c1-140
RCF_BULK_WEIGHT_PREP=0 python tools/step_prof.py fp32 6 2>&1 | tail -1
python tools/step_prof.py fp32 6 2>&1 | tail -1
RCF_BULK_WEIGHT_PREP=0 python tools/step_prof.py bf16 6 2>&1 | tail -1
python tools/step_prof.py bf16 6 2>&1 | tail -1
