mkdir -p gpurun_out/r3v
python -m pytest tests -m gpu -q > gpurun_out/r3v/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3v/gputests.log; tail -3 gpurun_out/r3v/gputests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3v/bench.json 2> gpurun_out/r3v/bench.err; cut -c1-300 gpurun_out/r3v/bench.json
PROF_ROWS=90 bash tools/prof_bench.sh r3v --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3v/prof.txt 2>&1; head -6 gpurun_out/r3v/prof.txt | cut -c1-160
bash tools/pmc_traffic.sh > gpurun_out/r3v/traffic.txt 2>&1; cat gpurun_out/r3v/traffic.txt
python tools/layer_table.py fp32 > gpurun_out/r3v/layers_fp32.txt 2>&1; grep "====\|family totals" gpurun_out/r3v/layers_fp32.txt
