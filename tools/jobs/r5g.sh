mkdir -p gpurun_out/r5g
RCF_WGRAD_BIG=1 bash tools/pmc_any.sh wbig igemm_wgrad_h2t tools/pmc_conv.py wgrad > gpurun_out/r5g/pmc_wgrad_big.txt 2>&1; cat gpurun_out/r5g/pmc_wgrad_big.txt | cut -c1-400
RCF_WGRAD_BIG=0 bash tools/pmc_any.sh wsmall igemm_wgrad_h2t tools/pmc_conv.py wgrad > gpurun_out/r5g/pmc_wgrad_small.txt 2>&1; cat gpurun_out/r5g/pmc_wgrad_small.txt | cut -c1-400
