mkdir -p gpurun_out/r3n
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "conv or train_step or commuted" > gpurun_out/r3n/tests.log 2>&1; tail -4 gpurun_out/r3n/tests.log
timeout 600 python tools/bench_h2p.py 16 > gpurun_out/r3n/h2p.txt 2>&1; cut -c1-330 gpurun_out/r3n/h2p.txt
python tools/layer_table.py fp32 > gpurun_out/r3n/layers.txt 2>&1; grep "====\|family totals" gpurun_out/r3n/layers.txt
