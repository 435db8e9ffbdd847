#!/bin/bash
# usage: tools/prof_any.sh <tag> <calls-divisor> <script> [args...] -> rocprofv3 --kernel-trace --stats of `python3 script args`,
# top kernels with time per call group (total / divisor)
tag=$1; div=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
setsid rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python3 $R/"$1" "${@:2}" > $R/gpurun_out/prof_$tag.log 2>&1 &
pid=$!
( sleep ${PROF_LIMIT:-300}; kill -KILL -- -$pid 2>/dev/null ) &
wd=$!
wait $pid
kill $wd 2>/dev/null
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_$tag/p_kernel_stats.csv")))
n = $div
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"sum of kernel time {tot / n / 1e6:.3f} ms per unit ({n} units)")
for r in rows[:${PROF_ROWS:-16}]:
    print(f'{float(r["TotalDurationNs"]) / n / 1e6:8.3f} ms {int(r["Calls"]) / n:7.1f} calls {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Name"][:110]}')
PY
