mkdir -p gpurun_out/r3p
python -m pytest tests -m gpu -q > gpurun_out/r3p/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3p/gputests.log; tail -3 gpurun_out/r3p/gputests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err; cut -c1-300 gpurun_out/r3p/bench.json
python tools/layer_table.py fp32 > gpurun_out/r3p/layers_fp32.txt 2>&1; grep "====\|family totals" gpurun_out/r3p/layers_fp32.txt
python tools/layer_table.py bf16 > gpurun_out/r3p/layers_bf16.txt 2>&1; grep "====\|family totals" gpurun_out/r3p/layers_bf16.txt
PROF_ROWS=80 bash tools/prof_bench.sh r3p --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3p/prof.txt 2>&1; head -8 gpurun_out/r3p/prof.txt | cut -c1-160
bash tools/pmc_traffic.sh > gpurun_out/r3p/traffic.txt 2>&1; cat gpurun_out/r3p/traffic.txt
