mkdir -p gpurun_out/r3k
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "h2p" > gpurun_out/r3k/tests.log 2>&1; tail -15 gpurun_out/r3k/tests.log
timeout 600 python tools/bench_h2p.py 16 > gpurun_out/r3k/h2p.txt 2>&1; cat gpurun_out/r3k/h2p.txt
