mkdir -p gpurun_out/r4s
python tools/fuzz_conv.py 150 3 > gpurun_out/r4s/fuzz3.txt 2>&1; tail -3 gpurun_out/r4s/fuzz3.txt | cut -c1-250
python tools/fuzz_conv.py 150 11 > gpurun_out/r4s/fuzz11.txt 2>&1; tail -3 gpurun_out/r4s/fuzz11.txt | cut -c1-250
python tools/fuzz_shapes.py > gpurun_out/r4s/fuzz_shapes.txt 2>&1; tail -4 gpurun_out/r4s/fuzz_shapes.txt | cut -c1-250
