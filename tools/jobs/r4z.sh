mkdir -p gpurun_out/r4z
python -m pytest tests -m gpu -q > gpurun_out/r4z/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4z/gputests.log; tail -4 gpurun_out/r4z/gputests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4z/bench.json 2> gpurun_out/r4z/bench.err; cut -c1-250 gpurun_out/r4z/bench.json; tail -2 gpurun_out/r4z/bench.err
PROF_ROWS=90 bash tools/prof_bench.sh r4z --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4z/prof.txt 2>&1; head -3 gpurun_out/r4z/prof.txt | cut -c1-160
