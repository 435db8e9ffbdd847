python tools/absmax_trace.py fp32 2>&1 | tail -30 | cut -c1-200
python tools/step_prof.py fp32 6 2>&1 | tail -1
