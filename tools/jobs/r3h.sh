mkdir -p gpurun_out/r3h
python tools/bench_epilogue.py > gpurun_out/r3h/epilogue.txt 2>&1
PROF_ROWS=28 bash tools/prof_any.sh r3h_crf_smooth 4 tools/crf_prof.py smooth 5 > gpurun_out/r3h/crf_smooth.txt 2>&1
PROF_ROWS=28 bash tools/prof_any.sh r3h_crf_noise 4 tools/crf_prof.py noise 5 > gpurun_out/r3h/crf_noise.txt 2>&1
cat gpurun_out/r3h/epilogue.txt; cat gpurun_out/r3h/crf_smooth.txt; cat gpurun_out/r3h/crf_noise.txt
