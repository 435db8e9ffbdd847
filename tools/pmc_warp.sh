#!/bin/bash
# HBM-side fetch/write bytes of the warp kernels (tools/bench_warp.py): separate FETCH_SIZE / WRITE_SIZE passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcw_$c -- python3 $R/tools/bench_warp.py > $R/gpurun_out/pmcw_$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$R/gpurun_out/pmcw_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"]][c] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]][c] += 1
for k, d in agg.items():
    f = d.get("FETCH_SIZE", 0) / max(cnt[k]["FETCH_SIZE"], 1); w = d.get("WRITE_SIZE", 0) / max(cnt[k]["WRITE_SIZE"], 1)
    print(f"{k[:70]:70s} launches {max(cnt[k].values()):4d} FETCH_SIZE {f:12.1f} KiB WRITE_SIZE {w:12.1f} KiB -> (2F+W) {(2*f+w)*1024/1e6:9.1f} MB, (F+W) {(f+w)*1024/1e6:9.1f} MB")
PY
