#!/bin/bash
# usage: tools/pmc_warp.sh   -> counter passes over tools/bench_warp.py (big shape only), per-kernel averages
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_warp_$i -- python3 $R/tools/bench_warp.py big > $R/gpurun_out/pmc_warp_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in range(1, 7):
    fs = glob.glob("$R/gpurun_out/pmc_warp_%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs:
        print("pass", i, "no output:", open("$R/gpurun_out/pmc_warp_%d.log" % i).read()[-300:])
    for f in fs:
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in agg.items():
            if "warp" in k:
                print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
