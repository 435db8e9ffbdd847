mkdir -p gpurun_out/r3l
python -m pytest tests -m gpu -q > gpurun_out/r3l/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3l/gputests.log; tail -3 gpurun_out/r3l/gputests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3l/bench.json 2> gpurun_out/r3l/bench.err; cut -c1-600 gpurun_out/r3l/bench.json
PROF_ROWS=70 bash tools/prof_bench.sh r3l --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3l/prof.txt 2>&1; head -12 gpurun_out/r3l/prof.txt | cut -c1-200
bash tools/pmc_traffic.sh > gpurun_out/r3l/traffic.txt 2>&1; cat gpurun_out/r3l/traffic.txt
bash tools/pmc_crf.sh smooth > gpurun_out/r3l/crf_smooth.txt 2>&1; head -14 gpurun_out/r3l/crf_smooth.txt | cut -c1-220
bash tools/pmc_crf.sh noise > gpurun_out/r3l/crf_noise.txt 2>&1; head -14 gpurun_out/r3l/crf_noise.txt | cut -c1-220
