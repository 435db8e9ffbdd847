#!/bin/bash
# usage: tools/pmc_fetch.sh <tag> <script> [args...]: FETCH_SIZE / WRITE_SIZE passes (separate, MI355X_MICROARCH.md HBM section) over a
# script; prints per kernel: launches, average KiB of each counter, corrected bytes (2*FETCH + WRITE) * 1024 per launch.  Environment passes through.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_${tag}_$c -- python3 $R/"$1" "${@:2}" > $R/gpurun_out/pmcf_${tag}_$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$R/gpurun_out/pmcf_${tag}_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"]][c] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]][c] += 1
rows = []
for k, d in agg.items():
    f = d.get("FETCH_SIZE", 0) / max(cnt[k]["FETCH_SIZE"], 1); w = d.get("WRITE_SIZE", 0) / max(cnt[k]["WRITE_SIZE"], 1)
    rows.append(((2 * f + w) * 1024 * max(cnt[k].values()), k, max(cnt[k].values()), f, w))
rows.sort(reverse=True)
print("kernel | launches | FETCH_SIZE avg KiB | WRITE_SIZE avg KiB | corrected MB per launch")
for tot, k, n, f, w in rows[:${PMC_ROWS:-12}]:
    print(f"{k[:110]} | {n} | {f:.1f} | {w:.1f} | {(2*f+w)*1024/1e6:.1f}")
PY
