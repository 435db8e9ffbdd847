mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q > gpurun_out/r3a/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/gputests.log
python tools/layer_table.py fp32 > gpurun_out/r3a/layers_fp32.txt 2>&1
python tools/layer_table.py bf16 > gpurun_out/r3a/layers_bf16.txt 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
tail -3 gpurun_out/r3a/gputests.log
