mkdir -p gpurun_out/r3b
timeout 600 python tools/bench_h2p.py 16 > gpurun_out/r3b/h2p.txt 2>&1
tail -12 gpurun_out/r3b/h2p.txt
