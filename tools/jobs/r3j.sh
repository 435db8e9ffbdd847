mkdir -p gpurun_out/r3j
python -m pytest tests/test_crf_gpu.py tests/test_stage2_gpu.py tests/test_abi_gpu.py -x -q > gpurun_out/r3j/tests.log 2>&1; tail -3 gpurun_out/r3j/tests.log
python tools/time_crf.py > gpurun_out/r3j/time_crf.txt 2>&1; cat gpurun_out/r3j/time_crf.txt
