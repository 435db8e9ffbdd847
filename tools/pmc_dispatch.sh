#!/bin/bash
# usage: tools/pmc_dispatch.sh <fp32|bf16> <fwd|dgrad|wgrad>: FETCH_SIZE / WRITE_SIZE (separate passes) of the SECOND launch of each
# shape of tools/pmc_shapes.py, next to the launch's algorithmic bytes -> fabric traffic ratio per layer shape
prec=$1; which=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${prec}_${which}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcd_${tag}_$c -- python3 $R/tools/pmc_shapes.py $prec $which > $R/gpurun_out/pmcd_${tag}_$c.log 2>&1
done
python3 $R/tools/pmc_shapes.py $prec $which --list > /tmp/shapes_$tag.txt
python3 - <<PY
import csv, glob
shapes = [l.strip().split("|") for l in open("/tmp/shapes_$tag.txt")]
pat = {"fp32": ("igemm_conv_x3_kernel", "conv_h2p_kernel", "conv_h2s_kernel", "igemm_wgrad"), "bf16": ("conv_bf16_kernel", "wgrad_bf16")}["$prec"]
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob("$R/gpurun_out/pmcd_${tag}_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and any(p in r["Kernel_Name"] for p in pat):
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    val[c] = rows
n = len(val["FETCH_SIZE"])
print(f"# $prec $which: per launch (second of two), (2*FETCH_SIZE + WRITE_SIZE) KiB -> MB, against x + y + w bytes; {n} matching dispatches for {len(shapes)} shapes")
if n == 2 * len(shapes) and len(val["WRITE_SIZE"]) == n:
    for i, (name, alg, flops) in enumerate(shapes):
        f, w = val["FETCH_SIZE"][2 * i + 1], val["WRITE_SIZE"][2 * i + 1]
        mb = (2 * f[2] + w[2]) * 1024 / 1e6
        print(f"{name:40s} {f[1][26:70]:44s} fetch {2*f[2]*1024/1e6:8.1f} MB write {w[2]*1024/1e6:7.1f} MB total {mb:8.1f} MB | algorithmic {float(alg)/1e6:7.1f} MB | ratio {mb/(float(alg)/1e6):5.2f}")
else:
    for r in val["FETCH_SIZE"]:
        print(r)
PY
