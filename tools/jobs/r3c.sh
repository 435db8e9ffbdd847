mkdir -p gpurun_out/r3c
RCF_H2P=1 bash tools/pmc_any.sh r3c_h2p conv_h2p tools/pmc_conv.py fwd > gpurun_out/r3c/pmc_h2p.txt 2>&1
RCF_H2P=0 bash tools/pmc_any.sh r3c_x3 igemm_conv_x3 tools/pmc_conv.py fwd > gpurun_out/r3c/pmc_x3.txt 2>&1
cat gpurun_out/r3c/pmc_h2p.txt gpurun_out/r3c/pmc_x3.txt
