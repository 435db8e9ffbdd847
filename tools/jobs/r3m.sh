mkdir -p gpurun_out/r3m
timeout 600 python tools/bench_tiles.py > gpurun_out/r3m/tiles.txt 2>&1; cat gpurun_out/r3m/tiles.txt
