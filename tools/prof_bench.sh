#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench.py args...]  -> gpurun_out/prof_<tag>/ (rocprofv3 --kernel-trace --stats, csv)
# The profiled process runs in its own process group under a watchdog: a hang at exit cannot eat the GPU call.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
setsid rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python3 $R/bench.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1 &
pid=$!
( sleep ${PROF_LIMIT:-420}; kill -KILL -- -$pid 2>/dev/null ) &
wd=$!
wait $pid
kill $wd 2>/dev/null
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -${PROF_ROWS:-30} "$f" | cut -c1-220
grep '"metric"' $R/gpurun_out/prof_$tag.log | head -1 | cut -c1-300
