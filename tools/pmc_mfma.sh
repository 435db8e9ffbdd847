#!/bin/bash
# Matrix-pipe utilisation, effective clock and L2 behaviour of every kernel of the training steps as bench.py runs them
# (VERDICT round 5, item 2): two rocprofv3 counter passes (SQ + GRBM; TCC), each with --kernel-trace only, the program directly
# behind `--`.  usage: tools/pmc_mfma.sh [tag]   ->  gpurun_out/pmc_mfma_<tag>.txt / .json   (tools/pmc_mfma_report.py)
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export RCF_BENCH_PRIMING=1
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma_${tag}_$i -- python3 $R/bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-stage2 > $R/gpurun_out/pmc_mfma_${tag}_$i.log 2>&1
  echo "pass $i rc $?"
done
python3 $R/tools/pmc_mfma_report.py $R/gpurun_out/pmc_mfma_${tag}_1 $R/gpurun_out/pmc_mfma_${tag}_2 $R/gpurun_out/pmc_mfma_${tag}
# the raw CSVs are large: keep the report only
rm -rf $R/gpurun_out/pmc_mfma_${tag}_1 $R/gpurun_out/pmc_mfma_${tag}_2
