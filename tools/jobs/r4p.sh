mkdir -p gpurun_out/r4p
RCF_X3_BIG=1 python -m pytest tests/test_kernels_gpu.py -q -k "h2p or conv_fp16_pairs or column_tile" 2>&1 | tail -3
RCF_X3_BIG=0 RCF_H2P=0 python tools/ab_korder.py fp32 > gpurun_out/r4p/x3big0.txt 2>&1
RCF_X3_BIG=1 RCF_H2P=0 python tools/ab_korder.py fp32 > gpurun_out/r4p/x3big1.txt 2>&1
RCF_X3_BIG=0 python tools/ab_korder.py fp32 > gpurun_out/r4p/h2p.txt 2>&1
for f in x3big0 x3big1 h2p; do echo "== $f"; grep -v amdgpu gpurun_out/r4p/$f.txt | cut -c1-40,62-105,156-215; done
python tools/ab_step_knob.py x3_big 3 6 > gpurun_out/r4p/ab_step.txt 2>&1; grep "fp32 step" gpurun_out/r4p/ab_step.txt
