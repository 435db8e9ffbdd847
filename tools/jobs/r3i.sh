mkdir -p gpurun_out/r3i
timeout 300 python tools/bench_epilogue.py > gpurun_out/r3i/epilogue.txt 2>&1
cat gpurun_out/r3i/epilogue.txt
