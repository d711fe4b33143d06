#!/bin/bash
# timing variants of k_bilinear_bwd_wide (diag build) + two PMC passes over the shipped kernel
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/wide_probe; mkdir -p $O
timeout 600 python tools/probes/wide_bwd_probe.py "$@" 2>&1 | tail -24 | tee $O/variants.txt
cd /tmp
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
         "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf $GRAFT_REPO_ROOT/$O/pmc_$n
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$n -o pmc -- python $GRAFT_REPO_ROOT/tools/probes/wide_bwd_probe.py > $GRAFT_REPO_ROOT/$O/pmc_$n.log 2>&1
  echo "pmc $n rc=$?"
  python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $GRAFT_REPO_ROOT/$O/pmc_$n k_bilinear_bwd_wide | tee -a $GRAFT_REPO_ROOT/$O/pmc.txt
  find $GRAFT_REPO_ROOT/$O/pmc_$n -name "*.csv" -size +2M -delete
done
