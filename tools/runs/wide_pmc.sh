#!/bin/bash
# PMC passes over the two fused FiBiNET kernels inside the model's step (tools/bench_leg.py fibinet)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/wide_pmc; mkdir -p $O; rm -f $O/pmc.txt
cd /tmp
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
         "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf $O/pmc_$n
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_leg.py fibinet --steps 10 --warmup 3 --warmup-seconds 0.2 --repeats 1 --no-graph > $O/pmc_$n.log 2>&1
  echo "pmc $n rc=$?"
  python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $O/pmc_$n k_bilinear_bwd_wide k_bilinear_fwd_wide | tee -a $O/pmc.txt
  rm -rf $O/pmc_$n
done
