python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-200
RCF_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-300
