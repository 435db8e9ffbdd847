mkdir -p gpurun_out/r3w
python bench.py --steps 20 --warmup 5 > gpurun_out/r3w/bench.json 2> gpurun_out/r3w/bench.err; cut -c1-250 gpurun_out/r3w/bench.json; tail -3 gpurun_out/r3w/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3w/bench.json'))
print(d['roofline']['family'], d['roofline']['frac'], d['roofline'].get('frac_one_stream'), d['roofline']['traffic'], d['bf16_frames_per_s'])
PY
timeout 600 python -m pytest tests/test_dist_gpu.py -q -k "bench" 2>&1 | tail -2
