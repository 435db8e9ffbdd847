#!/bin/bash
# usage: tools/pmc_run.sh <tag> <python script + args...>   -> gpurun_out/pmc_<tag>_<n>/ csv per counter pass
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/"$1" "${@:2}" > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in range(1, 5):
    fs = glob.glob("$R/gpurun_out/pmc_${tag}_%d/**/*counter_collection.csv" % i, recursive=True)
    for f in fs:
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in agg.items():
            if "igemm" in k or "wgrad" in k or "conv_bf16" in k:
                print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
