mkdir -p gpurun_out/r3f
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "h2p or fused_bn or commuted or train_step" > gpurun_out/r3f/tests.log 2>&1; tail -5 gpurun_out/r3f/tests.log
python tools/layer_table.py fp32 > gpurun_out/r3f/layers_h2p.txt 2>&1
grep "====\|family totals" gpurun_out/r3f/*.txt
awk '/second stream: False/{p=1} p' gpurun_out/r3f/layers_h2p.txt | grep " k3 " | grep "conv_dgrad_wide" | cut -c1-150
