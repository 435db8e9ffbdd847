mkdir -p gpurun_out/r4f
python -m pytest tests -m gpu -q > gpurun_out/r4f/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4f/gputests.log; tail -4 gpurun_out/r4f/gputests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err; cut -c1-250 gpurun_out/r4f/bench.json; tail -2 gpurun_out/r4f/bench.err
PROF_ROWS=45 bash tools/prof_step.sh f32 fp32 4 > gpurun_out/r4f/prof_fp32.txt 2>&1; head -14 gpurun_out/r4f/prof_fp32.txt | cut -c1-160
PROF_ROWS=45 bash tools/prof_step.sh b16 bf16 4 > gpurun_out/r4f/prof_bf16.txt 2>&1; head -14 gpurun_out/r4f/prof_bf16.txt | cut -c1-160
bash tools/pmc_traffic.sh > gpurun_out/r4f/pmc_traffic.log 2>&1; tail -10 gpurun_out/r4f/pmc_traffic.log
