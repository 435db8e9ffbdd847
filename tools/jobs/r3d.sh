mkdir -p gpurun_out/r3d
python -m pytest tests/test_kernels_gpu.py -x -q -k "h2p or fp16_pairs or region or fused_bn" > gpurun_out/r3d/tests.log 2>&1; tail -5 gpurun_out/r3d/tests.log
python tools/layer_table.py fp32 > gpurun_out/r3d/layers_h2p.txt 2>&1
RCF_H2P=0 python tools/layer_table.py fp32 > gpurun_out/r3d/layers_x3.txt 2>&1
python tools/layer_table.py fp32 > gpurun_out/r3d/layers_h2p_b.txt 2>&1
grep "====\|family totals" gpurun_out/r3d/*.txt
