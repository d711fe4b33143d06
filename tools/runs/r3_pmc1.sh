#!/bin/bash
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_pmc1
mkdir -p $O
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_]*\|TCP_[A-Z_0-9]*\|TCC_[A-Z_0-9]*" | sort -u > $O/counters.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3"; do
i=$((i+1))
timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o pmc -- python $GRAFT_REPO_ROOT/tools/tower_bench.py --nx 1 --iters 20 > $O/p$i.log 2>&1
echo "pass $i rc=$?"
python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $O/p$i k_mlp_train k_mlp_wgrad k_mlp_reduce > $O/p$i.txt 2>&1
done
rm -rf $O/p1 $O/p2 $O/p3
