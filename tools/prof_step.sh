#!/bin/bash
# usage: tools/prof_step.sh <tag> <fp32|bf16> [steps] [pairs] [schedule_field=value ...]  -> gpurun_out/prof_<tag>/ (rocprofv3 --kernel-trace --stats, csv) of
# tools/step_prof.py; prints the top of the kernel statistics with per-step milliseconds
tag=$1; prec=$2; steps=${3:-4}; pairs=${4:-8}; shift 4 2>/dev/null || shift $#
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
setsid rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python3 $R/tools/step_prof.py $prec $steps $pairs "$@" > $R/gpurun_out/prof_$tag.log 2>&1 &
pid=$!
( sleep ${PROF_LIMIT:-420}; kill -KILL -- -$pid 2>/dev/null ) &
wd=$!
wait $pid
kill $wd 2>/dev/null
tail -1 $R/gpurun_out/prof_$tag.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_$tag/p_kernel_stats.csv")))
n = $steps + 3
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"sum of kernel time {tot / n / 1e6:.2f} ms/step over {n} steps")
for r in rows[:${PROF_ROWS:-32}]:
    print(f'{float(r["TotalDurationNs"]) / n / 1e6:8.3f} ms/step {int(r["Calls"]) / n:7.1f} calls/step {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Name"][:130]}')
PY
