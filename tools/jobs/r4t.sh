mkdir -p gpurun_out/r4t
RCF_X3_BIG=1 python -m pytest tests/test_kernels_gpu.py -q -k "h2p or conv_fp16_pairs or column_tile or fused_bn" 2>&1 | tail -3
RCF_X3_BIG=0 RCF_H2P=0 python tools/ab_korder.py fp32 > gpurun_out/r4t/x3big0.txt 2>&1
RCF_X3_BIG=1 RCF_H2P=0 python tools/ab_korder.py fp32 > gpurun_out/r4t/x3big1.txt 2>&1
for f in x3big0 x3big1; do echo "== $f"; grep -v amdgpu gpurun_out/r4t/$f.txt | cut -c1-40,62-105,156-215; done
RCF_X3_BIG=0 python tools/step_prof.py fp32 6 2>&1 | tail -2
RCF_X3_BIG=1 python tools/step_prof.py fp32 6 2>&1 | tail -2
RCF_X3_BIG=1 RCF_X3_BIG_MIN_K=2048 python tools/step_prof.py fp32 6 2>&1 | tail -2
