mkdir -p gpurun_out/r4q
python -m pytest tests -m gpu -q > gpurun_out/r4q/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4q/gputests.log; tail -4 gpurun_out/r4q/gputests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4q/bench.json 2> gpurun_out/r4q/bench.err; cut -c1-250 gpurun_out/r4q/bench.json; tail -2 gpurun_out/r4q/bench.err
PROF_ROWS=90 bash tools/prof_bench.sh r4q --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4q/prof.txt 2>&1; head -3 gpurun_out/r4q/prof.txt | cut -c1-160
PROF_ROWS=50 bash tools/prof_step.sh f32 fp32 4 > gpurun_out/r4q/prof_fp32.txt 2>&1; head -4 gpurun_out/r4q/prof_fp32.txt | cut -c1-160
PROF_ROWS=50 bash tools/prof_step.sh b16 bf16 4 > gpurun_out/r4q/prof_bf16.txt 2>&1; head -4 gpurun_out/r4q/prof_bf16.txt | cut -c1-160
bash tools/pmc_traffic.sh > gpurun_out/r4q/pmc_traffic.log 2>&1; tail -9 gpurun_out/r4q/pmc_traffic.log
python tools/layer_table.py fp32 > gpurun_out/r4q/layers_fp32.txt 2>&1; grep "====\|family totals" gpurun_out/r4q/layers_fp32.txt | cut -c1-300
