mkdir -p gpurun_out/r3y
python tools/ab_wgrad_xcd.py > gpurun_out/r3y/ab_wgrad_xcd.txt 2>&1; cat gpurun_out/r3y/ab_wgrad_xcd.txt | cut -c1-330
python tools/ab_step_knob.py wgrad_xcd 3 6 > gpurun_out/r3y/ab_step.txt 2>&1; tail -4 gpurun_out/r3y/ab_step.txt
RCF_AB_ONE=0 bash tools/pmc_fetch.sh xcd0 tools/ab_wgrad_xcd.py > gpurun_out/r3y/pmc_xcd0.txt 2>&1; head -8 gpurun_out/r3y/pmc_xcd0.txt | cut -c1-200
RCF_AB_ONE=1 bash tools/pmc_fetch.sh xcd1 tools/ab_wgrad_xcd.py > gpurun_out/r3y/pmc_xcd1.txt 2>&1; head -8 gpurun_out/r3y/pmc_xcd1.txt | cut -c1-200
