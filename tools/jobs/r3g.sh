mkdir -p gpurun_out/r3g
python -m pytest tests -m gpu -q > gpurun_out/r3g/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3g/gputests.log
tail -30 gpurun_out/r3g/gputests.log
