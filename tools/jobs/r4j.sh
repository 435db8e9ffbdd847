mkdir -p gpurun_out/r4j
python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py -q -x 2>&1 | tail -2
python tools/ab_step_knob.py colmap 3 6 > gpurun_out/r4j/ab_step_colmap.txt 2>&1; tail -4 gpurun_out/r4j/ab_step_colmap.txt
for p in fp32 bf16; do bash tools/pmc_dispatch.sh $p dgrad > gpurun_out/r4j/pmcd_${p}_dgrad.txt 2>&1; head -6 gpurun_out/r4j/pmcd_${p}_dgrad.txt | cut -c1-260; done
RCF_COLMAP=0 python tools/ab_korder.py fp32 > gpurun_out/r4j/dgrad_colmap0.txt 2>&1; RCF_COLMAP=1 python tools/ab_korder.py fp32 > gpurun_out/r4j/dgrad_colmap1.txt 2>&1
grep "dh[23] coarse" gpurun_out/r4j/dgrad_colmap0.txt gpurun_out/r4j/dgrad_colmap1.txt | cut -c1-330
RCF_COLMAP=0 python tools/ab_korder.py bf16 > gpurun_out/r4j/dgrad16_colmap0.txt 2>&1; RCF_COLMAP=1 python tools/ab_korder.py bf16 > gpurun_out/r4j/dgrad16_colmap1.txt 2>&1
grep "dh[23] coarse" gpurun_out/r4j/dgrad16_colmap0.txt gpurun_out/r4j/dgrad16_colmap1.txt | cut -c1-330
