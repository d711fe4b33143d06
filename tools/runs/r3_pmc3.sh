#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_pmc3
mkdir -p $O
cd /tmp
P1="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"
P2="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_EA0_RDREQ_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
i=$((i+1))
timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/a$i -o pmc -- python $GRAFT_REPO_ROOT/tools/tower_bench.py --nx 1 --iters 20 > $O/a$i.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $O/a$i k_mlp_train k_mlp_wgrad > $O/standalone_p$i.txt 2>&1
timeout 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/b$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --steps 32 --warmup 8 --repeats 1 --warmup-seconds 0 --no-graph > $O/b$i.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $O/b$i k_mlp_train k_mlp_wgrad k_embed_fwd k_embed_apply_sorted > $O/instep_p$i.txt 2>&1
done
rm -rf $O/a1 $O/a2 $O/b1 $O/b2
