export TMPDIR=/tmp; cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tls
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tls -o tls -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 12 --no-cpu-baseline --force-parallel --kernel-iters 2 > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/tls
