mkdir -p gpurun_out/r3x
python -m pytest tests -m gpu -q > gpurun_out/r3x/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3x/gputests.log; tail -3 gpurun_out/r3x/gputests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3x/bench.json 2> gpurun_out/r3x/bench.err; cut -c1-250 gpurun_out/r3x/bench.json; tail -2 gpurun_out/r3x/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3x/bench.json'))
print(d['roofline']['family'], d['roofline']['frac'], d['roofline'].get('frac_one_stream'), d['bf16_frames_per_s'], d['bf16_step']['roofline']['frac'], d['config'])
PY
PROF_ROWS=90 bash tools/prof_bench.sh r3x --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3x/prof.txt 2>&1; head -5 gpurun_out/r3x/prof.txt | cut -c1-160
