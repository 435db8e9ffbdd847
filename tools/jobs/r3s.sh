mkdir -p gpurun_out/r3s
RCF_OVERLAP_WGRAD=0 PROF_ROWS=45 bash tools/prof_step.sh r3s_fp32 fp32 4 > gpurun_out/r3s/fp32.txt 2>&1; cat gpurun_out/r3s/fp32.txt | cut -c1-230
RCF_OVERLAP_WGRAD=0 PROF_ROWS=40 bash tools/prof_step.sh r3s_bf16 bf16 4 > gpurun_out/r3s/bf16.txt 2>&1; cat gpurun_out/r3s/bf16.txt | cut -c1-230
