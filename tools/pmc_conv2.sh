#!/bin/bash
# usage: tools/pmc_conv2.sh <tag> <fwd|dgrad|wgrad> [tile]  -> SQ wave-cycle breakdown + MFMA busy of one bf16 conv shape (two counter passes)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/tools/pmc_conv_bf16.py "$@" > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in range(1, 3):
    for f in glob.glob("$R/gpurun_out/pmc_${tag}_%d/**/*counter_collection.csv" % i, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in agg.items():
            if "conv_bf16" in k or "wgrad_bf16" in k:
                print("${tag}", k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
