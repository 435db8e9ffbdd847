mkdir -p gpurun_out/r3z
python -m pytest tests/test_kernels_gpu.py -q -k "xcd_mapping or wgrad" 2>&1 | tail -3
for w in fwd dgrad wgrad; do for p in fp32 bf16; do bash tools/pmc_dispatch.sh $p $w > gpurun_out/r3z/pmcd_${p}_$w.txt 2>&1; cat gpurun_out/r3z/pmcd_${p}_$w.txt | cut -c1-260; done; done
