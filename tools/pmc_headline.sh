#!/bin/bash
# random vs all-zero operands for the shipped headline conv kernels: MFMA busy, effective clock, L2 (tools/pmc_headline.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in random zero; do
  i=0
  for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
              "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_head_${mode}_$i -- python3 $R/tools/pmc_headline.py $mode > $R/gpurun_out/pmc_head_${mode}_$i.log 2>&1
    echo "$mode pass $i rc $?"
  done
  python3 $R/tools/pmc_mfma_report.py $R/gpurun_out/pmc_head_${mode}_1 $R/gpurun_out/pmc_head_${mode}_2 $R/gpurun_out/pmc_headline_${mode} > /dev/null
  rm -rf $R/gpurun_out/pmc_head_${mode}_1 $R/gpurun_out/pmc_head_${mode}_2
done
# the same two runs un-profiled, kernel trace only (wall-clock of the un-instrumented launches)
for mode in random zero; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_head_${mode} -- python3 $R/tools/pmc_headline.py $mode 10 > $R/gpurun_out/kt_head_${mode}.log 2>&1
  f=$(find $R/gpurun_out/kt_head_${mode} -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/kt_headline_${mode}_stats.csv
  rm -rf $R/gpurun_out/kt_head_${mode}
done
