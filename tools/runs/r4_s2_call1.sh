#!/bin/bash
# round 4, second session: (1) SQ counters of the step engine's kernels (three --pmc passes, eager engine steps),
# (2) FiBiNET launch order on the current sources, (3) DeepFM bench line as a check of the rebuilt library
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_s2_1
mkdir -p $O
cd $GRAFT_REPO_ROOT
B="--no-other-configs --no-cpu-baseline --steps 200 --warmup 20"
timeout 300 python bench.py $B > $O/bench_engine.json 2> $O/bench_engine.err
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3"; do
i=$((i+1))
timeout 240 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_one_model.py DeepFM 24 > $O/p$i.log 2>&1
echo "pass $i rc=$?"
python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $O/p$i k_embed_tower_train k_mlp_wgrad k_mlp_reduce k_embed_apply_sorted > $O/p$i.txt 2>&1
done
rm -rf $O/p1 $O/p2 $O/p3
cat $O/p1.txt $O/p2.txt $O/p3.txt > $O/tower_sq_counters.txt
rm -rf /tmp/prof_f
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_f -o fib -- python $GRAFT_REPO_ROOT/tools/prof_one_model.py FiBiNET 8 > $O/prof_fib.log 2>&1
t=$(find /tmp/prof_f -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/step_profile.py $t 2 4 > $O/fibinet_step_kernel_budget.txt 2>&1
python $GRAFT_REPO_ROOT/tools/step_sequence.py $t 2 > $O/fibinet_step_launch_order.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_s2_1/bench_engine.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d["value"], d["roofline"]["frac"])
PY
cat $O/tower_sq_counters.txt | cut -c1-400
head -45 $O/fibinet_step_launch_order.txt | cut -c1-160
