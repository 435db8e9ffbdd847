mkdir -p gpurun_out/r4k
python -m pytest tests -m gpu -q > gpurun_out/r4k/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4k/gputests.log; tail -4 gpurun_out/r4k/gputests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4k/bench.json 2> gpurun_out/r4k/bench.err; cut -c1-250 gpurun_out/r4k/bench.json; tail -2 gpurun_out/r4k/bench.err
PROF_ROWS=90 bash tools/prof_bench.sh r4k --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4k/prof.txt 2>&1; head -3 gpurun_out/r4k/prof.txt | cut -c1-160
PROF_ROWS=50 bash tools/prof_step.sh f32 fp32 4 > gpurun_out/r4k/prof_fp32.txt 2>&1; head -4 gpurun_out/r4k/prof_fp32.txt | cut -c1-160
PROF_ROWS=50 bash tools/prof_step.sh b16 bf16 4 > gpurun_out/r4k/prof_bf16.txt 2>&1; head -4 gpurun_out/r4k/prof_bf16.txt | cut -c1-160
bash tools/pmc_traffic.sh > gpurun_out/r4k/pmc_traffic.log 2>&1; tail -9 gpurun_out/r4k/pmc_traffic.log
python tools/layer_table.py fp32 > gpurun_out/r4k/layers_fp32.txt 2>&1; grep "====\|family totals" gpurun_out/r4k/layers_fp32.txt | cut -c1-300
python tools/layer_table.py bf16 > gpurun_out/r4k/layers_bf16.txt 2>&1; grep "====\|family totals" gpurun_out/r4k/layers_bf16.txt | cut -c1-300
python tools/stability_run.py both 250 > gpurun_out/r4k/stability.txt 2>&1; tail -6 gpurun_out/r4k/stability.txt | cut -c1-250
