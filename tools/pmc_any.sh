#!/bin/bash
# usage: tools/pmc_any.sh <tag> <kernel-name-substring> <script> [args...]  -> counter passes, per-kernel averages of the kernels
# whose name contains the substring.  Every pass runs under its own timeout (an unknown counter name must not hang the box).
tag=$1; filt=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN SQ_INSTS_FLAT" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
            "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/"$1" "${@:2}" > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in range(1, 6):
    fs = glob.glob("$R/gpurun_out/pmc_${tag}_%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs:
        print("pass", i, "no output:", open("$R/gpurun_out/pmc_${tag}_%d.log" % i).read()[-300:])
    for f in fs:
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:56]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in agg.items():
            if "$filt" in k:
                print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
