#!/bin/bash
# HBM traffic of the CRF kernels, per case: tools/pmc_crf.sh <smooth|noise> [iters]  -> gpurun_out/pmc_crf_<case>.txt
# separate FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md, HBM section: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024)
kind=$1; iters=${2:-5}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_crf_${kind}_$c -- python3 $R/tools/crf_prof.py $kind $iters > $R/gpurun_out/pmc_crf_${kind}_$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
dur = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$R/gpurun_out/pmc_crf_${kind}_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            agg[r["Kernel_Name"]][c] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]][c] += 1
    for f in glob.glob("$R/gpurun_out/pmc_crf_${kind}_%s/**/*kernel_trace.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
calls = 4          # crf_prof.py runs the head 4 times on 8 frames
rows = []
for k, d in agg.items():
    n = max(cnt[k].values())
    f, w = d.get("FETCH_SIZE", 0) / max(cnt[k]["FETCH_SIZE"], 1), d.get("WRITE_SIZE", 0) / max(cnt[k]["WRITE_SIZE"], 1)
    us = sum(dur[k]) / max(len(dur[k]), 1) / 1e3
    rows.append(((2 * f + w) * 1024 * n, k, n, f, w, us))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows) / calls / 8
with open("$R/gpurun_out/pmc_crf_${kind}.txt", "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 tools/crf_prof.py ${kind} ${iters}   (8 frames of 480x854 per call, 4 calls)\n")
    o.write("# counter unit KiB; gfx950 correction: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; durations are those of the profiled passes\n")
    o.write(f"# corrected HBM bytes per frame, all kernels: {tot / 1e6:.1f} MB\n")
    o.write("kernel | launches per call | FETCH_SIZE avg KiB | WRITE_SIZE avg KiB | corrected HBM MB per launch | avg us | MB per frame\n")
    for t, k, n, f, w, us in rows[:24]:
        o.write(f"{k[:100]} | {n / calls:.1f} | {f:.1f} | {w:.1f} | {(2*f+w)*1024/1e6:.2f} | {us:.1f} | {t / calls / 8 / 1e6:.2f}\n")
print(open("$R/gpurun_out/pmc_crf_${kind}.txt").read())
PY
